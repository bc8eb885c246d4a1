"""Probe: pgpu_modexp (host arrays) with perfect-square moduli, split form on / off: wall time and kernel times of the
library's HIP-event timers (tools/, diagnostics only)."""
import sys, time, json, ctypes
sys.path.insert(0, ".")
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
from pailliercryptolib_amd.limbs import ints_to_limbs
pa.initialize(0)
L = _capi.lib()
L.pgpu_debug_set_hensel.argtypes = [ctypes.c_int]
k = json.load(open("tests/golden/iso_kat.json"))
p, q = int(k["p"], 16), int(k["q"], 16)
n = p * q
rng = np.random.default_rng(1)
for name, mod, W in (("n^2", n * n, 64), ("p^2", p * p, 32)):
    for count in (2048, 8192, 32768):
        base = np.frombuffer(rng.bytes(count * W * 8), dtype=np.uint64).reshape(count, W).copy()
        base[:, -1] >>= np.uint64(2)
        E = W // 2
        exp = np.frombuffer(rng.bytes(count * E * 8), dtype=np.uint64).reshape(count, E).copy()
        m = ints_to_limbs([mod], W)[0]
        for mode in (0, 1):
            L.pgpu_debug_set_hensel(mode)
            pa.mod_exp_limbs(base, exp, m, exp_bits=64 * E)
            _capi.check(L.pgpu_set_timing(1))
            t0 = time.perf_counter()
            out = pa.mod_exp_limbs(base, exp, m, exp_bits=64 * E)
            dt = time.perf_counter() - t0
            kinds = (ctypes.c_int * 64)()
            ms = (ctypes.c_double * 64)()
            got = L.pgpu_timing_collect(kinds, ms, 64)
            _capi.check(L.pgpu_set_timing(0))
            print(name, count, "hensel", mode, "wall ms", round(dt * 1e3, 2), "kernels", [round(ms[i], 2) for i in range(got)],
                  "hash", hash(out.tobytes()) & 0xffff, flush=True)
pa.terminate()
