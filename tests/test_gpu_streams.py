"""`_dev` entry points on different HIP streams must not share scratch memory (VERDICT r01 item 8 / ADVICE: the window
tables and the CRT hand-over buffer used to be process-global).  Two torch streams issue encrypt + CRT decrypt +
CT x PT + CT + CT on different data, interleaved and unsynchronised; every result is compared with the oracle
(encrypt) and with the plaintext algebra (decrypt of the homomorphic results)."""
import json
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_streams_do_not_share_workspaces(engine):
    import torch
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    pa = engine
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
    n = p * q
    nsq = n * n
    pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    N = 2048                                   # 4096 half-width instances per decrypt: kernels of both streams overlap
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    data, out = [], [None, None]
    for si in range(2):
        rng = random.Random(100 + si)
        m = [rng.randrange(n) for _ in range(N)]
        r = [rng.getrandbits(1024) for _ in range(N)]
        e = [rng.getrandbits(33) for _ in range(N)]
        data.append((m, r, e, T.to_device(ints_to_limbs(m, 32)), T.to_device(ints_to_limbs(r, 16)),
                     T.to_device(ints_to_limbs(e, 1))))
    torch.cuda.synchronize()
    for rep in range(3):                       # several rounds: later launches find warm, shared-or-not workspaces
        for si in range(2):
            m, r, e, d_m, d_r, d_e = data[si]
            with torch.cuda.stream(streams[si]):
                c = T.encrypt(pk, d_m, d_r, 1024)
                ce = T.mod_exp(c, d_e, nsq, 33)            # CT x PT
                s = T.mod_mul(ce, c, nsq)                  # CT + CT: (e + 1) * m
                out[si] = (c, T.decrypt(sk, c), T.decrypt(sk, s))
    torch.cuda.synchronize()
    for si in range(2):
        m, r, e = data[si][:3]
        c, d1, d2 = out[si]
        got_c = limbs_to_ints(T.to_host(c[:64]))
        assert got_c == opk.encrypt(m[:64], r[:64]), f"stream {si}: ciphertexts differ from the oracle"
        assert limbs_to_ints(T.to_host(d1)) == m, f"stream {si}: decrypt(encrypt(m)) != m"
        assert limbs_to_ints(T.to_host(d2)) == [(x + 1) * y % n for x, y in zip(e, m)], f"stream {si}: chain wrong"
