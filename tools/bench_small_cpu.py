"""CPU time of the reference's benchmark sizes (benchmark/bench_cryptography.cpp:10-19: 16 ... 2100 elements) with the
oracle's AVX512-IFMA restatement on all usable cores: the number to hold beside the GPU's small-batch latencies
(profiles/r02_ipcl_api_bench.txt).  Test infrastructure only.  (tools/, diagnostics)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from oracle import paillier_oracle as orc  # noqa: E402
from pailliercryptolib_amd.limbs import ints_to_limbs  # noqa: E402

k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
n = p * q
nw, pw = 32, 16
sk = orc.PrivateKey(n, p, q)
threads = min(c_oracle.lib().orc_max_threads(), c_oracle.usable_cpus())
c_oracle.set_threads(threads)
be = c_oracle.ifma_modexp_batch if c_oracle.ifma_lib() is not None else c_oracle.openssl_modexp_batch
args = [ints_to_limbs([v], pw)[0] for v in (sk.p, sk.q, sk.hp, sk.hq, sk.pinv)]
n_l, hs_l = ints_to_limbs([n], nw)[0], ints_to_limbs([hs], 2 * nw)[0]
rng = np.random.default_rng(1)
print(f"# {threads} threads, backend {'ifma' if c_oracle.ifma_lib() is not None else 'openssl'}; us per call, best of 5")
print(f"{'op':10s} {'batch':>6s} {'us/call':>12s}")
for size in (16, 64, 128, 256, 512, 1024, 2048, 2100):
    m = np.frombuffer(rng.bytes(size * nw * 8), dtype=np.uint64).reshape(size, nw).copy()
    m[:, -1] &= np.uint64((1 << 62) - 1)
    r = np.frombuffer(rng.bytes(size * pw * 8), dtype=np.uint64).reshape(size, pw).copy()
    best_e = best_d = 1e30
    for _ in range(5):
        t0 = time.perf_counter(); c = c_oracle.paillier_encrypt_with(be, n_l, hs_l, m, r); best_e = min(best_e, time.perf_counter() - t0)
        t0 = time.perf_counter(); d = c_oracle.paillier_decrypt_crt_with(be, *args, c); best_d = min(best_d, time.perf_counter() - t0)
    assert np.array_equal(d, m)
    print(f"{'Encrypt':10s} {size:6d} {best_e * 1e6:12.1f}")
    print(f"{'Decrypt':10s} {size:6d} {best_d * 1e6:12.1f}")
