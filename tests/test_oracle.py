"""CPU suite: pins the oracle against the reference's own known-answer vectors
(CryptoTest.ISO_IEC_18033_6_ComplianceTest, reference test/test_cryptography.cpp:99-241) and the
committed seeded fixtures.  No GPU, no /root/reference access."""
import json
import os

import pytest

from oracle import paillier_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    return {key: (int(v, 16) if isinstance(v, str) and v.startswith("0x") else v) for key, v in k.items()}


def test_iso_kat_encrypt_add_decrypt(kat):
    p, q = kat["p"], kat["q"]
    n = p * q
    assert n.bit_length() == 2048
    pk = orc.PublicKey(n, n.bit_length())            # non-DJN key: test_cryptography.cpp:118
    sk = orc.PrivateKey(n, p, q)
    num = kat["num_values"]
    m = [kat["m0"]] * num
    r = [kat["r0"]] * num
    m[1], r[1] = kat["m1"], kat["r1"]                # test_cryptography.cpp:198-203
    ct = pk.encrypt(m, r)
    assert ct[0] == kat["c1"] and ct[1] == kat["c2"]             # :221-225
    s = orc.ct_add(ct[0:1], ct[1:2], n * n)
    assert s[0] == kat["c1c2"]                                   # :227-233
    assert sk.decrypt(s)[0] == kat["m1m2"]                       # :235-240
    assert sk.decrypt(ct) == m                                   # :215-219
    assert sk.decrypt(ct, crt=False) == m                        # decryptRAW, pri_key.cpp:92-111


def test_private_key_orders_p_q(kat):
    sk = orc.PrivateKey(kat["p"] * kat["q"], kat["p"], kat["q"])
    assert sk.p < sk.q and sk.p == kat["q"]          # the ISO key is given with q < p (Q2)


def test_bench_constants_consistent(kat):
    n = kat["p"] * kat["q"]
    assert kat["bench_r"] == kat["r0"]               # R_BN == r0 (bench_cryptography.cpp:37-46)
    assert kat["bench_hs"] < n * n


def test_seeded_fixtures_roundtrip():
    data = json.load(open(os.path.join(GOLD, "seeded_vectors.json")))
    assert len(data["cases"]) == 6
    for case in data["cases"]:
        p, q = int(case["p"], 16), int(case["q"], 16)
        n = p * q
        pk = orc.PublicKey(n, case["bits"])
        if case["djn"]:
            pk.set_djn(int(case["hs"], 16))
        sk = orc.PrivateKey(n, p, q)
        m = [int(v, 16) for v in case["m"]]
        r = [int(v, 16) for v in case["r"]]
        c = [int(v, 16) for v in case["c"]]
        assert pk.encrypt(m, r) == c
        assert sk.decrypt(c) == m
        assert orc.ct_add(c, c[::-1], n * n) == [int(v, 16) for v in case["add"]]
        e = [int(v, 16) for v in case["mul_exp"]]
        assert orc.ct_mul_pt(c, e, n * n) == [int(v, 16) for v in case["mul"]]
        # homomorphic properties: Dec(c1*c2) = m1+m2, Dec(c^e) = m*e
        assert sk.decrypt(orc.ct_add(c, c[::-1], n * n)) == [(a + b) % n for a, b in zip(m, m[::-1])]
        assert sk.decrypt(orc.ct_mul_pt(c, e, n * n)) == [a * b % n for a, b in zip(m, e)]


def test_error_behaviour():
    pk = orc.PublicKey(15, 4)
    with pytest.raises(RuntimeError):
        pk.encrypt([], [])                       # pub_key.cpp:116
    with pytest.raises(RuntimeError):
        orc.mod_exp_batch([1, 2], [1], [3, 3])   # mod_exp.cpp:452-454
    with pytest.raises(RuntimeError):
        orc.ct_add([1, 2, 3], [1, 2], 35)        # ciphertext.cpp:37-38
    with pytest.raises(RuntimeError):
        orc.PrivateKey(15, 3, 3)


def test_limb_helpers():
    x = (1 << 200) + 12345
    assert orc.from_limbs(orc.to_limbs(x, 4)) == x
    with pytest.raises(ValueError):
        orc.to_limbs(1 << 64, 1)
