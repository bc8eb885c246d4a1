"""Quick device-resident throughput probe of the modexp kernel (not the contract bench)."""
import ctypes, random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
from pailliercryptolib_amd.limbs import ints_to_limbs

def run(mod_bits, exp_bits, count, shared_exp, reps=3):
    rng = random.Random(1)
    mod = rng.getrandbits(mod_bits) | (1 << (mod_bits - 1)) | 1
    W = mod_bits // 64; E = (exp_bits + 63) // 64
    base = np.frombuffer(np.random.default_rng(1).bytes(count * W * 8), dtype=np.uint64).reshape(count, W).copy()
    base[:, -1] &= (1 << 62) - 1
    ne = 1 if shared_exp else count
    exp = np.frombuffer(np.random.default_rng(2).bytes(ne * E * 8), dtype=np.uint64).reshape(ne, E).copy()
    if exp_bits % 64: exp[:, -1] &= (1 << (exp_bits % 64)) - 1
    d_base = torch.from_numpy(base.view(np.int64)).cuda()
    d_exp = torch.from_numpy(exp.view(np.int64)).cuda()
    d_out = torch.empty((count, W), dtype=torch.int64, device="cuda")
    h_mod = ints_to_limbs([mod], W)[0]
    L = _capi.lib()
    s = torch.cuda.current_stream().cuda_stream
    def call():
        _capi.check(L.pgpu_modexp_dev(d_base.data_ptr(), W, d_exp.data_ptr(), 0 if shared_exp else E, E, exp_bits,
                                      h_mod.ctypes.data_as(ctypes.c_void_p), W, d_out.data_ptr(), count, ctypes.c_void_p(s)))
    call(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); call(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    s32 = mod_bits // 32
    M = 2 * s32 * s32 + s32
    w = 5 if exp_bits >= 128 else 2
    Nmm = exp_bits + (exp_bits + w - 1) // w + (1 << w)
    mac = M * Nmm * count
    print(f"mod={mod_bits} exp={exp_bits} n={count} shared_exp={shared_exp}: {best*1e3:.2f} ms  {count/best:,.0f} modexp/s  "
          f"{mac/best/1e12:.2f} T MAC32/s ({mac/best/39.32e12*100:.1f}% of 39.32T)")

pa.initialize()
run(4096, 1024, 8192, False)
run(4096, 1024, 16384, False)
run(4096, 1024, 32768, False)
run(2048, 1024, 16384, True)
run(2048, 1024, 32768, True)
run(4096, 2048, 8192, True)
run(1024, 512, 16384, True)
run(6144, 1536, 8192, False)
run(4096, 32, 65536, False)
run(2048, 1024, 65536, True)
run(4096, 1024, 65536, False)
