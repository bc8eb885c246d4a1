"""Probe: table entries of the fixed-window path per kernel form (tools/, diagnostics only)."""
import sys, random, ctypes
sys.path.insert(0, ".")
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
pa.initialize()
L = _capi.lib()
L.pgpu_debug_set_row_source.argtypes = [ctypes.c_int]
rng = random.Random(5)
for rs in (0, 1):
    L.pgpu_debug_set_row_source(rs)
    for bits in (512, 1024):
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        W = bits // 64
        base = [rng.randrange(mod) for _ in range(4)]
        for ebits in (8, 20):
            for e in (0, 1, 2, 3, 4, 5, 6, 7, 9, 31):
                out = pa.mod_exp_limbs(ints_to_limbs(base, W), ints_to_limbs([e] * 4, 1), ints_to_limbs([mod], W)[0], exp_bits=ebits)
                got = limbs_to_ints(out)
                ok = got == [pow(b, e, mod) for b in base]
                if not ok:
                    # which power did we get?
                    which = [next((k for k in range(64) if pow(b, k, mod) == g), None) for b, g in zip(base, got)]
                    print("rs", rs, "bits", bits, "exp_bits", ebits, "e", e, "WRONG; got powers", which, flush=True)
print("done")
pa.terminate()
