"""GPU parity at BASELINE.json's full per-GPU sizes through size-independent properties
(encrypt -> decrypt round trip, homomorphic identities) plus bit-exact oracle checks on samples.
Everything stays resident in HBM (torch tensors + the *_dev C-ABI entry points)."""
import json
import os
import random

import numpy as np
import pytest

from oracle import paillier_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rand_rows(rng, count, words, top_mask=None):
    a = np.frombuffer(rng.bytes(count * words * 8), dtype=np.uint64).reshape(count, words).copy()
    if top_mask is not None:
        a[:, -1] &= np.uint64(top_mask)
    return a


@pytest.fixture(scope="module")
def iso():
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    return int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)


def test_config2_and_3_encrypt_decrypt_8192(engine, iso):
    """configs[1] + configs[2]: k=2048, batch 8192 DJN encrypt then CRT decrypt."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import limbs_to_ints
    p, q, hs = iso
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = np.random.default_rng(2)
    N = 8192
    m = rand_rows(rng, N, 32, (1 << 62) - 1)
    r = rand_rows(rng, N, 16)
    d_m, d_r = T.to_device(m), T.to_device(r)
    d_c = T.encrypt(pk, d_m, d_r)
    d_out = T.decrypt(sk, d_c)
    assert torch.equal(d_out, d_m)                                   # round trip, all 8192
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    idx = [0, 1, 4095, 8191] + random.Random(1).sample(range(N), 8)
    c = T.to_host(d_c[idx])
    assert limbs_to_ints(c) == opk.encrypt(limbs_to_ints(m[idx]), limbs_to_ints(r[idx]))
    # the non-DJN leg (r^n, 2048-bit exponent) on the same batch
    pk2 = engine.PublicKey(n, 2048)
    d_r2 = T.to_device(rand_rows(rng, N, 32, (1 << 62) - 1))
    d_c2 = T.encrypt(pk2, d_m, d_r2)
    assert torch.equal(T.decrypt(sk, d_c2), d_m)
    opk2 = orc.PublicKey(n, 2048)
    assert limbs_to_ints(T.to_host(d_c2[:2])) == opk2.encrypt(limbs_to_ints(m[:2]), limbs_to_ints(T.to_host(d_r2[:2])))


def test_config4_3072_bit_shard(engine):
    """configs[3]: k=3072 (beyond the reference's 2048-bit cap), one GPU's 8192-element shard of the
    65536 batch: DJN encrypt + CRT decrypt; n^2 is 6144 bits (Geo<16,14>), p^2 3072 bits (Geo<8,14>)."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import limbs_to_ints
    case = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"]
            if c["bits"] == 3072 and c["djn"]][0]
    p, q, hs = int(case["p"], 16), int(case["q"], 16), int(case["hs"], 16)
    n = p * q
    pk, sk = engine.PublicKey(n, 3072, hs=hs), engine.PrivateKey(p, q)
    rng = np.random.default_rng(4)
    N = 8192
    m = rand_rows(rng, N, 48, (1 << 62) - 1)
    r = rand_rows(rng, N, 24)                       # 1536-bit r
    d_m = T.to_device(m)
    d_c = T.encrypt(pk, d_m, T.to_device(r))
    assert torch.equal(T.decrypt(sk, d_c), d_m)
    opk = orc.PublicKey(n, 3072)
    opk.set_djn(hs)
    idx = [0, 8191, 777]
    assert limbs_to_ints(T.to_host(d_c[idx])) == opk.encrypt(limbs_to_ints(m[idx]), limbs_to_ints(r[idx]))


def test_config5_add_and_mul_131072(engine, iso):
    """configs[4]: k=2048, one GPU's 131072-element shard of the 1 M batch: CT+CT (modmul mod n^2)
    and CT*PT with u32 / u64 exponents (short-exponent modexp)."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    p, q, hs = iso
    n = p * q
    nsq = n * n
    rng = np.random.default_rng(5)
    N = 131072
    a = rand_rows(rng, N, 64, (1 << 62) - 1)        # uniform values < 2^4094 < n^2 (pure modmul, config 5(i))
    b = rand_rows(rng, N, 64, (1 << 62) - 1)
    d_a, d_b = T.to_device(a), T.to_device(b)
    d_s = T.mod_mul(d_a, d_b, nsq)
    idx = [0, 1, N - 1] + random.Random(5).sample(range(N), 13)
    ai, bi = limbs_to_ints(a[idx]), limbs_to_ints(b[idx])
    assert limbs_to_ints(T.to_host(d_s[idx])) == [x * y % nsq for x, y in zip(ai, bi)]
    # commutativity and scalar broadcast over the whole batch
    assert torch.equal(T.mod_mul(d_b, d_a, nsq), d_s)
    d_one = T.to_device(ints_to_limbs([1], 64))
    assert torch.equal(T.mod_mul(d_a, d_one, nsq), T.mod_mul(d_a, T.to_device(np.tile(ints_to_limbs([1], 64), (N, 1))), nsq))
    # CT*PT: a^e with 32-bit and 64-bit exponents; (a^e1)^... identity: a^(e1) * a^(e2) == a^(e1+e2)
    e1 = rand_rows(rng, N, 1, (1 << 32) - 1)
    e2 = rand_rows(rng, N, 1, (1 << 32) - 1)
    d_p1 = T.mod_exp(d_a, T.to_device(e1), nsq, exp_bits=32)
    d_p2 = T.mod_exp(d_a, T.to_device(e2), nsq, exp_bits=32)
    d_p12 = T.mod_exp(d_a, T.to_device(e1 + e2), nsq, exp_bits=33)
    assert torch.equal(T.mod_mul(d_p1, d_p2, nsq), d_p12)
    assert limbs_to_ints(T.to_host(d_p1[idx])) == [pow(x, int(e), nsq) for x, e in zip(ai, e1[idx, 0])]
    e64 = rand_rows(rng, 4096, 1)
    d_p64 = T.mod_exp(d_a[:4096], T.to_device(e64), nsq)
    assert limbs_to_ints(T.to_host(d_p64[:4])) == [pow(x, int(e), nsq) for x, e in zip(limbs_to_ints(a[:4]), e64[:4, 0])]
    # real ciphertexts: Dec(c1 * c2) = m1 + m2 on a slice
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    m1 = rand_rows(rng, 2048, 32, (1 << 61) - 1)
    m2 = rand_rows(rng, 2048, 32, (1 << 61) - 1)
    c1 = T.encrypt(pk, T.to_device(m1), T.to_device(rand_rows(rng, 2048, 16)))
    c2 = T.encrypt(pk, T.to_device(m2), T.to_device(rand_rows(rng, 2048, 16)))
    got = limbs_to_ints(T.to_host(T.decrypt(sk, T.mod_mul(c1, c2, nsq))))
    assert got == [(x + y) % n for x, y in zip(limbs_to_ints(m1), limbs_to_ints(m2))]


@pytest.mark.parametrize("mod_bits,count", [(1024, 140000), (2048, 70000), (4096, 33000)])
def test_wide_geometries_large_batches(engine, mod_bits, count):
    """Batches large enough that the launcher picks the wide split (G/2 lanes x 18 limbs) of the
    modulus context: Geo<2,18>, Geo<4,18>, Geo<8,18>.  Short exponents keep it fast; checked against
    the oracle on samples and through a^(e1+e2) == a^e1 * a^e2 on the whole batch."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import limbs_to_ints
    rng = np.random.default_rng(mod_bits)
    pyrng = random.Random(mod_bits)
    mod = pyrng.getrandbits(mod_bits) | (1 << (mod_bits - 1)) | 1
    W = mod_bits // 64
    a = rand_rows(rng, count, W, (1 << 62) - 1)
    e1 = rand_rows(rng, count, 1, (1 << 20) - 1)
    e2 = rand_rows(rng, count, 1, (1 << 20) - 1)
    d_a = T.to_device(a)
    p1 = T.mod_exp(d_a, T.to_device(e1), mod, exp_bits=20)
    p2 = T.mod_exp(d_a, T.to_device(e2), mod, exp_bits=20)
    p12 = T.mod_exp(d_a, T.to_device(e1 + e2), mod, exp_bits=21)
    assert torch.equal(T.mod_mul(p1, p2, mod), p12)
    idx = [0, count - 1] + pyrng.sample(range(count), 6)
    assert limbs_to_ints(T.to_host(p1[idx])) == [pow(x % mod, int(e), mod) for x, e in zip(limbs_to_ints(a[idx]), e1[idx, 0])]


def test_wide_split_in_fused_encrypt_and_decrypt(engine, iso):
    """16384 ciphertexts: the fixed-base encrypt launches as Geo<8,18> and the two-context decrypt as
    Geo<4,18> (wide splits); same round-trip / oracle checks as at 8192."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import limbs_to_ints
    p, q, hs = iso
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = np.random.default_rng(16384)
    N = 16384
    m = rand_rows(rng, N, 32, (1 << 62) - 1)
    r = rand_rows(rng, N, 16)
    d_m = T.to_device(m)
    d_c = T.encrypt(pk, d_m, T.to_device(r))
    assert torch.equal(T.decrypt(sk, d_c), d_m)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    idx = [0, N - 1, 12345]
    assert limbs_to_ints(T.to_host(d_c[idx])) == opk.encrypt(limbs_to_ints(m[idx]), limbs_to_ints(r[idx]))
    # and the narrow split on a prefix gives the same ciphertexts
    assert torch.equal(T.encrypt(pk, d_m[:4096].contiguous(), T.to_device(r[:4096])), d_c[:4096])


@pytest.mark.parametrize("count", [8191, 8201, 16391])
def test_ragged_batches_on_the_wide_split(engine, iso, count):
    """Batch sizes that are not multiples of the instances per wavefront, large enough for the wide lane
    split (>= 1024 wavefronts) and, for decrypt, for parity waves (every wavefront serves one of p^2 / q^2):
    the padded tail groups must neither corrupt nor skip elements."""
    import torch
    from pailliercryptolib_amd import torch_ops as T
    from pailliercryptolib_amd.limbs import limbs_to_ints
    p, q, hs = iso
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = np.random.default_rng(count)
    m = rand_rows(rng, count, 32, (1 << 62) - 1)
    r = rand_rows(rng, count, 16)
    d_m = T.to_device(m)
    d_c = T.encrypt(pk, d_m, T.to_device(r))
    assert torch.equal(T.decrypt(sk, d_c), d_m)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    idx = [0, count - 1, count - 2, count - 9, count - 17]
    assert limbs_to_ints(T.to_host(d_c[idx])) == opk.encrypt(limbs_to_ints(m[idx]), limbs_to_ints(r[idx]))
