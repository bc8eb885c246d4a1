import sys, time, random, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import torch_ops as T
from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
pa.initialize()
print("init ok", flush=True)
rng = random.Random(3)
for bits, ebits, count in ((1024, 512, 64), (1024, 512, 16384), (1024, 1024, 16384), (1024, 512, 65536)):
    mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    W = bits // 64
    base = [rng.randrange(mod) for _ in range(64)] * (count // 64)
    e = [rng.getrandbits(ebits)]
    d_b = T.to_device(ints_to_limbs(base, W)); d_e = T.to_device(ints_to_limbs(e, (ebits + 63) // 64))
    print("launch", bits, ebits, count, flush=True)
    t0 = time.perf_counter()
    out = T.mod_exp(d_b, d_e, mod, ebits)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = limbs_to_ints(T.to_host(out[:64]))
    print("done", round(dt * 1e3, 2), "ms ok=", got == [pow(b, e[0], mod) for b in base[:64]], flush=True)
