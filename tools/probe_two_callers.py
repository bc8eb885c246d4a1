"""Two synchronous callers of pgpu_paillier_encrypt + pgpu_paillier_decrypt_crt on host arrays of their own (pageable or
pinned): per-call wall times of each thread, and the library's kernel timeline (pgpu_timing_collect_trace).
usage: python tools/probe_two_callers.py [pageable|pinned] [callers] [rounds]   (tools/, diagnostics only)"""
import ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
variant = sys.argv[1] if len(sys.argv) > 1 else "pageable"
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
B = 8192
pa.initialize(0)
L = _capi.lib()
k = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
pk, sk = pa.PublicKey(p * q, 2048, hs=hs), pa.PrivateKey(p, q)
rng = np.random.default_rng(1)
m = np.frombuffer(rng.bytes(B * 256), dtype=np.uint64).reshape(B, 32).copy()
m[:, -1] &= np.uint64((1 << 62) - 1)
r = np.frombuffer(rng.bytes(B * 128), dtype=np.uint64).reshape(B, 16).copy()
ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
held = []


def pin(shape):
    nbytes = int(np.prod(shape)) * 8
    pp = ctypes.c_void_p()
    _capi.check(L.pgpu_host_alloc(nbytes, ctypes.byref(pp)))
    held.append(pp)
    return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(pp.value), dtype=np.uint64).reshape(shape)


bufs = []
for _ in range(ncall):
    if variant == "pageable":
        bufs.append((m.copy(), r.copy(), np.empty((B, 64), dtype=np.uint64), np.empty((B, 32), dtype=np.uint64)))
    else:
        a, b2 = pin((B, 32)), pin((B, 16))
        a[:], b2[:] = m, r
        bufs.append((a, b2, pin((B, 64)), pin((B, 32))))
bar = threading.Barrier(ncall + 1)
log = [[] for _ in range(ncall)]
T0 = [0.0]


def caller(kk):
    mm, rr, cc, dd = bufs[kk]
    for it in range(reps + 1):
        if it == 1:
            bar.wait()
        t0 = time.perf_counter()
        _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(mm), 32, 32, ptr(rr), 16, 16, 1024, ptr(cc), B))
        t1 = time.perf_counter()
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(cc), ptr(dd), B))
        t2 = time.perf_counter()
        if it:
            log[kk].append((t0 - T0[0], t1 - t0, t2 - t1))


th = [threading.Thread(target=caller, args=(kk,)) for kk in range(ncall)]
for t in th:
    t.start()
time.sleep(0.3)
_capi.check(L.pgpu_set_timing(1))
T0[0] = time.perf_counter()
bar.wait()
for t in th:
    t.join()
wall = time.perf_counter() - T0[0]
print(f"{variant}, {ncall} callers x {reps} rounds: wall {wall * 1e3:.2f} ms, {wall / (reps * ncall) * 1e3:.3f} ms per encrypt+decrypt, "
      f"{3 * B * reps * ncall / wall / 1e6:.3f} M modexps/s; results ok: {all(np.array_equal(b[3], m) for b in bufs)}")
for kk in range(ncall):
    print(f" caller {kk}: " + "  ".join(f"@{s * 1e3:6.2f} enc {e * 1e3:5.2f} dec {d * 1e3:5.2f}" for s, e, d in log[kk]))
mx = 256
kinds, forms, lanes = (ctypes.c_int * mx)(), (ctypes.c_int * mx)(), (ctypes.c_int * mx)()
st, ms = (ctypes.c_double * mx)(), (ctypes.c_double * mx)()
n = L.pgpu_timing_collect_trace(kinds, forms, lanes, st, ms, mx)
if n > 0:
    base = min(st[i] for i in range(n))
    print(" kernels (kind form lane start_ms dur_ms):")
    for i in sorted(range(n), key=lambda i: st[i]):
        print(f"  {kinds[i]:2d} {forms[i]:3d} {lanes[i]:3d} {st[i] - base:8.3f} {ms[i]:7.3f}")
L.pgpu_set_timing(0)
for pp in held:
    L.pgpu_host_free(pp)
