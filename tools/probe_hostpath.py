"""Where the time of a host-buffer CT+CT call goes (diagnostics)."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize()
L = _capi.lib()
k = json.load(open(os.path.join(ROOT, "tests/golden/iso_kat.json")))
N = int(k["p"], 16) * int(k["q"], 16)
NSQ = N * N
rng = np.random.default_rng(7)
n = 131072
a = np.frombuffer(rng.bytes(n * 64 * 8), dtype=np.uint64).reshape(n, 64).copy(); a[:, -1] &= np.uint64((1 << 62) - 1)
b = a[::-1].copy()
mod = np.array([(NSQ >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(64)], dtype=np.uint64)
out = np.zeros_like(a)
p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
def t(fn, reps=5):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("pgpu_modmul, preallocated out: %.2f ms" % t(lambda: _capi.check(L.pgpu_modmul(p(a), p(b), 64, p(mod), 64, p(out), n))))
d = ctypes.c_void_p()
_capi.check(L.pgpu_dev_alloc(a.nbytes, ctypes.byref(d)))
print("copy_h2d 67 MB: %.2f ms" % t(lambda: _capi.check(L.pgpu_copy_h2d(d, p(a), a.nbytes))))
print("copy_d2h 67 MB: %.2f ms" % t(lambda: _capi.check(L.pgpu_copy_d2h(p(out), d, a.nbytes))))
tmp = np.empty_like(a)
print("host memcpy 67 MB (numpy): %.2f ms" % t(lambda: np.copyto(tmp, a)))
