"""CT x PT on resident ciphertexts (BASELINE configs[4] (ii) and the reference's bench_ops.cpp:138-149 shape): kernel time by
HIP events for the multi-lane forms (PGPU_PS_DECRYPT policy 0) and the one-lane product-scanning form of the n^2 domain
(policy 2, csrc/hensel_ps_n2.hpp), same inputs, outputs compared.  usage: bench_ctpt.py [count=1048576] [exp_bits=32] [reps=3]
(tools/, diagnostics only)"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pailliercryptolib_amd as pa  # noqa: E402
from pailliercryptolib_amd import _capi  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
e_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pa.initialize(0)
L = _capi.lib()
k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
pk = pa.PublicKey(p * q, 2048, hs=hs)
W, ew = 64, (e_bits + 63) // 64
rng = np.random.default_rng(5)
a = np.frombuffer(rng.bytes(count * W * 8), dtype=np.uint64).reshape(count, W).copy()
a[:, -1] &= np.uint64((1 << 60) - 1)
e = np.frombuffer(rng.bytes(count * ew * 8), dtype=np.uint64).reshape(count, ew).copy()
if e_bits % 64:
    e[:, -1] &= np.uint64((1 << (e_bits % 64)) - 1)
ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)


def op(fn, *args):
    h = ctypes.c_void_p()
    _capi.check(fn(*args, ctypes.byref(h)))
    return h


ha = op(L.pgpu_batch_upload, ptr(a), count, W, W)
he = op(L.pgpu_batch_upload, ptr(e), count, ew, ew)
one = op(L.pgpu_batch_upload, ptr(np.array([[1] + [0] * (W - 1)], dtype=np.uint64)), 1, W, W)
rows = op(L.pgpu_batch_ct_add, pk._h, ha, one)             # the bases as pair rows (device-produced operands)
_capi.check(L.pgpu_synchronize())
w = min(range(1, 6), key=lambda v: ((1 << v) - 2) + (e_bits + v - 1) // v)
nwin = (e_bits + w - 1) // w
nsq, nmul = w * (nwin - 1), (1 << w) - 2 + (nwin - 1)
outs = {}
for pol, name in ((0, "multi-lane (hensel_modexp_seq_kernel<4,18> by size)"), (2, "one-lane product scanning (hensel_modexp_ps_kernel<75,28>)")):
    L.pgpu_debug_set_ps_decrypt(pol)
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_modexp_n2_kernel_form(pk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    h = op(L.pgpu_batch_ct_mul, pk._h, rows, he, e_bits)    # warm-up (workspace, code object)
    _capi.check(L.pgpu_synchronize())
    L.pgpu_batch_destroy(h)
    _capi.check(L.pgpu_set_timing(1))
    hs_ = [op(L.pgpu_batch_ct_mul, pk._h, rows, he, e_bits) for _ in range(reps)]
    _capi.check(L.pgpu_synchronize())
    kinds, ms = (ctypes.c_int * 64)(), (ctypes.c_double * 64)()
    nrec = L.pgpu_timing_collect(kinds, ms, 64)
    _capi.check(L.pgpu_set_timing(0))
    t = [ms[i] for i in range(nrec) if kinds[i] == 1]
    got = np.empty((count, W), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(hs_[0], ptr(got)))
    outs[pol] = got
    for x in hs_:
        L.pgpu_batch_destroy(x)
    if split.value == 4:
        K = limbs.value
        sq = K * (K + 1) // 2 + K * K + 2 * K * (K - 1)
        mul = 3 * K * K + 2 * K * (K - 1)
    else:
        l2 = lanes.value * limbs.value if split.value == 2 else lanes.value // 2 * limbs.value
        g = lanes.value if split.value == 2 else lanes.value // 2
        sq = l2 * (l2 + g) // 2 + 3 * l2 * l2 if split.value == 2 else 4 * l2 * l2
        mul = 5 * l2 * l2 if split.value == 2 else 6 * l2 * l2
    macs = (nsq * sq + nmul * mul) * count
    best = min(t)
    print(f"{name}: form (split {split.value}, lanes {lanes.value}, limbs {limbs.value})  kernel ms {[round(v, 3) for v in t]}  "
          f"best {best:.3f} ms = {count / best / 1e3:.2f} M CTxPT/s; w = {w}: {nsq} squarings + {nmul} products per element, "
          f"{macs / count / 1e3:.0f} k MAC32 executed per element = {macs / (best * 1e-3) / 1e12:.2f} T MAC32/s = {macs / (best * 1e-3) / 39.32e12:.3f} of 39.32")
L.pgpu_debug_set_ps_decrypt(1)
print("outputs identical:", bool(np.array_equal(outs[0], outs[2])))
pa.terminate()
