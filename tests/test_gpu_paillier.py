"""GPU parity of the fused Paillier paths (encrypt, CRT decrypt, add, mul) through the C-ABI,
against the reference's ISO/IEC 18033-6 known-answer vectors and the committed seeded fixtures."""
import json
import os
import random

import pytest

from oracle import paillier_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    return {key: (int(v, 16) if isinstance(v, str) and v.startswith("0x") else v) for key, v in k.items()}


def test_iso_kat_through_gpu(engine, kat):
    """Mirror of CryptoTest.ISO_IEC_18033_6_ComplianceTest (test_cryptography.cpp:99-241)."""
    p, q = kat["p"], kat["q"]
    n = p * q
    pk = engine.PublicKey(n, n.bit_length())
    sk = engine.PrivateKey(p, q)
    num = kat["num_values"]
    m = [kat["m0"]] * num
    r = [kat["r0"]] * num
    m[1], r[1] = kat["m1"], kat["r1"]
    ct = pk.encrypt(m, r)
    assert ct[0] == kat["c1"]
    assert ct[1] == kat["c2"]
    assert sk.decrypt(ct) == m
    s = engine.mod_mul(ct[0:1], ct[1:2], n * n)          # CT + CT
    assert s[0] == kat["c1c2"]
    assert sk.decrypt(s)[0] == kat["m1m2"]


def test_bench_key_djn(engine, kat):
    """The benchmark configuration (bench_cryptography.cpp:73-95): DJN with injected hs, r."""
    p, q = kat["p"], kat["q"]
    n = p * q
    hs = kat["bench_hs"]
    pk = engine.PublicKey(n, 2048, hs=hs)
    sk = engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    count = 33
    m = [p - 1024 * i for i in range(count)]              # bench_cryptography.cpp:87
    r = [kat["bench_r"]] * count                          # full-width R_BN as the exponent (Q4)
    ct = pk.encrypt(m, r)
    assert ct == opk.encrypt(m, r)
    assert sk.decrypt(ct) == m


@pytest.mark.parametrize("fbw", [0, 4, 8, 12])
def test_djn_encrypt_generic_vs_fixed_base(engine, kat, fbw):
    """The DJN obfuscator hs^r through the generic kernel (w=0) and through fixed-base tables of
    several window widths must give the same bits, incl. r = 0, r = 1, short and full-width r."""
    from pailliercryptolib_amd import _capi
    p, q = kat["p"], kat["q"]
    n = p * q
    hs = kat["bench_hs"]
    rng = random.Random(fbw)
    r = [0, 1, (1 << 1024) - 1, kat["bench_r"], rng.getrandbits(7)] + [rng.getrandbits(1024) for _ in range(12)]
    m = [0, n - 1] + [rng.randrange(n) for _ in range(len(r) - 2)]
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    want = opk.encrypt(m, r)
    _capi.check(_capi.lib().pgpu_set_fixed_base_window(fbw))
    try:
        pk = engine.PublicKey(n, 2048, hs=hs)
        assert pk.encrypt(m[4:], r[4:]) == want[4:]          # table sized for 1024-bit r ...
        assert pk.encrypt(m, r) == want                      # ... and regrown for the 2047-bit R_BN
        assert pk.encrypt(m[:2], r[:2]) == want[:2]          # r in {0, 1}
    finally:
        _capi.check(_capi.lib().pgpu_set_fixed_base_window(13))   # the library default


def test_seeded_fixtures(engine):
    data = json.load(open(os.path.join(GOLD, "seeded_vectors.json")))
    for case in data["cases"]:
        p, q = int(case["p"], 16), int(case["q"], 16)
        n = p * q
        pk = engine.PublicKey(n, case["bits"], hs=int(case["hs"], 16) if case["djn"] else None)
        sk = engine.PrivateKey(p, q)
        m = [int(v, 16) for v in case["m"]]
        r = [int(v, 16) for v in case["r"]]
        c = [int(v, 16) for v in case["c"]]
        assert pk.encrypt(m, r) == c, (case["bits"], case["djn"])
        assert sk.decrypt(c) == m, (case["bits"], case["djn"])
        assert engine.mod_mul(c, c[::-1], n * n) == [int(v, 16) for v in case["add"]]
        e = [int(v, 16) for v in case["mul_exp"]]
        assert engine.mod_exp(c, e, n * n) == [int(v, 16) for v in case["mul"]]


@pytest.mark.parametrize("count", [1, 7, 8, 9, 2100])
def test_roundtrip_batch_sizes(engine, kat, count):
    """dec(enc(x)) == x for random u32 plaintexts (CryptoTest.CryptoTest, test_cryptography.cpp:67-97)
    at the reference's benchmark batch sizes incl. non-multiples of 8."""
    p, q = kat["p"], kat["q"]
    n = p * q
    rng = random.Random(count)
    pk = engine.PublicKey(n, 2048, hs=kat["bench_hs"])
    sk = engine.PrivateKey(p, q)
    m = [rng.getrandbits(32) for _ in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    ct = pk.encrypt(m, r)
    assert sk.decrypt(ct) == m
    # a sample of ciphertexts checked bit-exactly against the oracle
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(kat["bench_hs"])
    for i in sorted(set([0, count - 1, count // 2])):
        assert ct[i] == opk.encrypt([m[i]], [r[i]])[0]


def test_homomorphic_ops(engine, kat):
    """CT+CT, CT*PT incl. multiply-by-zero (test_ops.cpp:126-367)."""
    p, q = kat["p"], kat["q"]
    n = p * q
    nsq = n * n
    rng = random.Random(7)
    pk = engine.PublicKey(n, 2048, hs=kat["bench_hs"])
    sk = engine.PrivateKey(p, q)
    a = [rng.getrandbits(32) for _ in range(14)]
    b = [rng.getrandbits(32) for _ in range(14)]
    ca = pk.encrypt(a, [rng.getrandbits(1024) for _ in a])
    cb = pk.encrypt(b, [rng.getrandbits(1024) for _ in b])
    assert sk.decrypt(engine.mod_mul(ca, cb, nsq)) == [x + y for x, y in zip(a, b)]
    assert sk.decrypt(engine.mod_mul(ca, cb[:1], nsq)) == [x + b[0] for x in a]
    assert sk.decrypt(engine.mod_exp(ca, b, nsq)) == [x * y for x, y in zip(a, b)]
    assert sk.decrypt(engine.mod_exp(ca, [0] * 14, nsq)) == [0] * 14       # CtMultiplyZeroPtTest


def test_key_errors(engine):
    import pailliercryptolib_amd as pa
    with pytest.raises(pa._capi.PgpuError):
        engine.PrivateKey(7, 7)                        # p == q (pri_key.cpp:35)
    with pytest.raises(RuntimeError):
        engine.PublicKey(77, 7).encrypt([], [])        # empty PlainText (pub_key.cpp:116)
