"""Diagnostic: per-wave run time spread of the decrypt modexp kernel (tail effect analysis)."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi, torch_ops as T
pa.initialize()
L = _capi.lib()
k = json.load(open("tests/golden/iso_kat.json"))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
n = p * q
pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
rng = np.random.default_rng(1)
N = 8192
m = np.frombuffer(rng.bytes(N * 32 * 8), dtype=np.uint64).reshape(N, 32).copy(); m[:, -1] &= (1 << 62) - 1
r = np.frombuffer(rng.bytes(N * 16 * 8), dtype=np.uint64).reshape(N, 16).copy()
d_c = T.encrypt(pk, T.to_device(m), T.to_device(r))
T.decrypt(sk, d_c); torch.cuda.synchronize()
buf = torch.zeros((4096, 3), dtype=torch.int64, device="cuda")
L.pgpu_debug_set_wave_clocks.argtypes = [ctypes.c_void_p]
L.pgpu_debug_set_wave_clocks(buf.data_ptr())
T.decrypt(sk, d_c); torch.cuda.synchronize()
L.pgpu_debug_set_wave_clocks(None)
b = buf.cpu().numpy()
b = b[b[:, 1] != 0]
dur = (b[:, 1] - b[:, 0]).astype(np.float64)
t0 = b[:, 0].min()
print("waves", len(b), "duration ticks: min %.0f  median %.0f  mean %.0f  max %.0f  (max/mean %.3f)" % (dur.min(), np.median(dur), dur.mean(), dur.max(), dur.max() / dur.mean()))
print("start spread: %.0f ticks;  last end - first start: %.0f;  mean dur / total %.3f" % (b[:, 0].max() - t0, b[:, 1].max() - t0, dur.mean() / (b[:, 1].max() - t0)))
for x in range(8):
    sel = (b[:, 2] & 0xF) == x
    if sel.any():
        print(f"XCC {x}: waves {sel.sum():4d}  mean {dur[sel].mean():.0f}  max {dur[sel].max():.0f}  end-max {(b[sel,1].max()-t0):.0f}")
hw = (b[:, 2] >> 8) & 0xFFFFFFFF
wave_id, simd, pipe, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print("wave slot ids", np.unique(wave_id), "simd ids", np.unique(simd))
key = (b[:, 2] & 0xF) * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
import collections
groups = collections.defaultdict(list)
for i, kk in enumerate(key):
    groups[int(kk)].append(i)
sizes = collections.Counter(len(v) for v in groups.values())
print("waves per (xcc,se,sh,cu,simd):", dict(sizes))
pairs = [v for v in groups.values() if len(v) == 2]
d = np.array([[dur[v[0]], dur[v[1]]] for v in pairs])
print("pairs", len(pairs), "mean |diff| %.0f" % np.abs(d[:, 0] - d[:, 1]).mean(), " mean sum %.0f" % d.sum(1).mean())
for v in pairs[:6]:
    print("  pair (slot, cycles):", [(int(wave_id[i]), int(dur[i])) for i in v])
pct = np.percentile(dur, [1, 10, 50, 90, 99])
print("percentiles 1/10/50/90/99:", pct)
