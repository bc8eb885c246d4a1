"""Per-GPU throughput of the BASELINE.json configs other than the bench line (device-resident,
HIP-event timed, best of 3).  Writes gpurun_out/config_sweep.json; not the contract benchmark."""
import json, os, random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import torch_ops as T

pa.initialize()
gold = "tests/golden"
k = json.load(open(f"{gold}/iso_kat.json"))
P, Q, HS = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
N = P * Q


def rows(rng, count, words, mask=None):
    a = np.frombuffer(rng.bytes(count * words * 8), dtype=np.uint64).reshape(count, words).copy()
    if mask is not None:
        a[:, -1] &= np.uint64(mask)
    return a


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def mac32(mod_bits, e):
    s = mod_bits // 32; w = 5 if e >= 128 else 2
    return (2 * s * s + s) * (e + (e + w - 1) // w + (1 << w))

out = {}
rng = np.random.default_rng(7)
# config 2/3: k=2048, 8192
pk, sk = pa.PublicKey(N, 2048, hs=HS), pa.PrivateKey(P, Q)
dm, dr = T.to_device(rows(rng, 8192, 32, (1 << 62) - 1)), T.to_device(rows(rng, 8192, 16))
dc = T.encrypt(pk, dm, dr)
t_e, t_d = timed(lambda: T.encrypt(pk, dm, dr)), timed(lambda: T.decrypt(sk, dc))
out["config2_encrypt_djn_k2048_n8192"] = {"ms": t_e, "encrypts_per_s": 8192 / t_e * 1e3}
out["config3_decrypt_crt_k2048_n8192"] = {"ms": t_d, "decrypts_per_s": 8192 / t_d * 1e3, "modexps_per_s": 16384 / t_d * 1e3,
                                          "frac_of_39.32T": 2 * mac32(2048, 1024) * 8192 / (t_d * 1e-3) / 39.32e12}
pk2 = pa.PublicKey(N, 2048)
dr2 = T.to_device(rows(rng, 8192, 32, (1 << 62) - 1))
t = timed(lambda: T.encrypt(pk2, dm, dr2))
out["config2_encrypt_nondjn_k2048_n8192"] = {"ms": t, "encrypts_per_s": 8192 / t * 1e3,
                                             "frac_of_39.32T": mac32(4096, 2048) * 8192 / (t * 1e-3) / 39.32e12}
# config 4: k=3072 shard of 8192
case = [c for c in json.load(open(f"{gold}/seeded_vectors.json"))["cases"] if c["bits"] == 3072 and c["djn"]][0]
p3, q3, hs3 = int(case["p"], 16), int(case["q"], 16), int(case["hs"], 16)
pk3, sk3 = pa.PublicKey(p3 * q3, 3072, hs=hs3), pa.PrivateKey(p3, q3)
dm3, dr3 = T.to_device(rows(rng, 8192, 48, (1 << 62) - 1)), T.to_device(rows(rng, 8192, 24))
dc3 = T.encrypt(pk3, dm3, dr3)
t_e, t_d = timed(lambda: T.encrypt(pk3, dm3, dr3)), timed(lambda: T.decrypt(sk3, dc3))
out["config4_k3072_shard8192"] = {"encrypt_ms": t_e, "decrypt_ms": t_d, "enc_plus_dec_per_s": 8192 / (t_e + t_d) * 1e3,
                                  "decrypt_frac_of_39.32T": 2 * mac32(3072, 1536) * 8192 / (t_d * 1e-3) / 39.32e12}
# config 5: k=2048, 131072-element shard
NSQ = N * N
da, db = T.to_device(rows(rng, 131072, 64, (1 << 62) - 1)), T.to_device(rows(rng, 131072, 64, (1 << 62) - 1))
t = timed(lambda: T.mod_mul(da, db, NSQ))
out["config5_add_ctct_n131072"] = {"ms": t, "modmuls_per_s": 131072 / t * 1e3, "algorithmic_GBs": 131072 * 1536 / (t * 1e-3) / 1e9,
                                   "frac_of_39.32T": 2 * (2 * 128 * 128 + 128) * 131072 / (t * 1e-3) / 39.32e12}
de32 = T.to_device(rows(rng, 131072, 1, (1 << 32) - 1))
t = timed(lambda: T.mod_exp(da, de32, NSQ, exp_bits=32))
out["config5_mul_ctpt_u32_n131072"] = {"ms": t, "modexps_per_s": 131072 / t * 1e3, "frac_of_39.32T": mac32(4096, 32) * 131072 / (t * 1e-3) / 39.32e12}
de64 = T.to_device(rows(rng, 131072, 1))
t = timed(lambda: T.mod_exp(da, de64, NSQ, exp_bits=64))
out["config5_mul_ctpt_u64_n131072"] = {"ms": t, "modexps_per_s": 131072 / t * 1e3, "frac_of_39.32T": mac32(4096, 64) * 131072 / (t * 1e-3) / 39.32e12}
# ---- end-to-end from HOST buffers through the synchronous C-ABI entry points (H2D + kernels + D2H; flat
# uint64 arrays, i.e. without the BigNumber marshalling of the C++ layer) ----
def wall(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

import ctypes
from pailliercryptolib_amd import _capi
Lc = _capi.lib()
vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
hm, hr = rows(rng, 8192, 32, (1 << 62) - 1), rows(rng, 8192, 16)
hc, hout = np.zeros((8192, 64), dtype=np.uint64), np.zeros((8192, 32), dtype=np.uint64)   # caller-owned, reused
enc_h = lambda: _capi.check(Lc.pgpu_paillier_encrypt(pk._h, vp(hm), 32, 32, vp(hr), 16, 16, 1024, vp(hc), 8192))
dec_h = lambda: _capi.check(Lc.pgpu_paillier_decrypt_crt(sk._h, vp(hc), vp(hout), 8192))
t_e, t_d = wall(enc_h), wall(dec_h)
assert np.array_equal(hout, hm)
out["host_buffers_config2_3_k2048_n8192"] = {"encrypt_ms": t_e, "decrypt_ms": t_d,
                                             "modexps_per_s": 3 * 8192 / (t_e + t_d) * 1e3}
ha, hb = T.to_host(da), T.to_host(db)
hs_out = np.zeros_like(ha)
nsq_l = np.array([(NSQ >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(64)], dtype=np.uint64)
t = wall(lambda: _capi.check(Lc.pgpu_modmul(vp(ha), vp(hb), 64, vp(nsq_l), 64, vp(hs_out), 131072)))
out["host_buffers_config5_add_n131072"] = {"ms": t, "modmuls_per_s": 131072 / t * 1e3,
                                           "host_GBs": 131072 * 1536 / (t * 1e-3) / 1e9}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/config_sweep.json", "w"), indent=1)
print(json.dumps(out, indent=1))
