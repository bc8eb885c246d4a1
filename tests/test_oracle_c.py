"""CPU suite: the C oracle (oracle/modexp_oracle.c, the bench's cpu_baseline "port") against the
reference's ISO KAT and against CPython pow (oracle/paillier_oracle.py)."""
import json
import os
import random

import numpy as np
import pytest

from oracle import c_oracle
from oracle import paillier_oracle as orc
from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_c_modexp_vs_pow():
    rng = random.Random(11)
    for bits, ebits, cnt in ((1024, 512, 5), (2048, 1024, 4), (4096, 1024, 3), (4096, 33, 4), (6144, 100, 2)):
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        W, E = bits // 64, (ebits + 63) // 64
        base = [rng.getrandbits(bits) for _ in range(cnt)]   # may exceed the modulus: reduced
        exp = [rng.getrandbits(ebits) for _ in range(cnt)]
        exp[0] = 0
        got = limbs_to_ints(c_oracle.modexp_batch(ints_to_limbs(base, W), ints_to_limbs(exp, E),
                                                  ints_to_limbs([mod], W)[0]))
        assert got == [pow(b % mod, e, mod) for b, e in zip(base, exp)]


def test_c_modmul_vs_python():
    rng = random.Random(12)
    mod = rng.getrandbits(4096) | (1 << 4095) | 1
    a = [rng.randrange(mod) for _ in range(6)]
    b = [rng.randrange(mod) for _ in range(6)]
    got = limbs_to_ints(c_oracle.modmul_batch(ints_to_limbs(a, 64), ints_to_limbs(b, 64), ints_to_limbs([mod], 64)[0]))
    assert got == [x * y % mod for x, y in zip(a, b)]


def test_c_paillier_iso_kat():
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    g = {key: (int(v, 16) if isinstance(v, str) and v.startswith("0x") else v) for key, v in k.items()}
    p, q = sorted((g["p"], g["q"]))
    n = p * q
    sk = orc.PrivateKey(n, p, q)       # host-side key constants (hp, hq, pinv), pinned in test_oracle.py
    m = [g["m0"], g["m1"]]
    r = [g["r0"], g["r1"]]
    c = c_oracle.paillier_encrypt(ints_to_limbs([n], 32)[0], None, ints_to_limbs(m, 2), ints_to_limbs(r, 32))
    ci = limbs_to_ints(c)
    assert ci[0] == g["c1"] and ci[1] == g["c2"]
    dm = c_oracle.paillier_decrypt_crt(*(ints_to_limbs([v], 16)[0] for v in (p, q, sk.hp, sk.hq, sk.pinv)), c)
    assert limbs_to_ints(dm) == m
    # DJN leg with the benchmark's hs
    hs = g["bench_hs"]
    rr = [random.Random(3).getrandbits(1024) for _ in range(2)]
    c2 = c_oracle.paillier_encrypt(ints_to_limbs([n], 32)[0], ints_to_limbs([hs], 64)[0], ints_to_limbs(m, 2),
                                   ints_to_limbs(rr, 16))
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    assert limbs_to_ints(c2) == opk.encrypt(m, rr)


def test_openssl_leg_agrees_with_pow_and_c_port():
    """OpenSSL BN_mod_exp_mont (the oracle of the reference's own QAT tests,
    module/heqat/test/test_bnModExp.cpp:57-60) vs CPython pow vs the C port."""
    if c_oracle.openssl_lib() is None:
        pytest.skip("libcrypto not available at build time")
    rng = random.Random(13)
    for bits, ebits, cnt in ((1024, 512, 4), (2048, 1024, 4), (4096, 1024, 3)):
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        W, E = bits // 64, (ebits + 63) // 64
        base = [rng.getrandbits(bits) for _ in range(cnt)]
        exp = [rng.getrandbits(ebits) for _ in range(cnt)]
        b, e, m = ints_to_limbs(base, W), ints_to_limbs(exp, E), ints_to_limbs([mod], W)[0]
        got = limbs_to_ints(c_oracle.openssl_modexp_batch(b, e, m))
        assert got == [pow(x % mod, y, mod) for x, y in zip(base, exp)]
        assert got == limbs_to_ints(c_oracle.modexp_batch(b, e, m))


def test_ifma_leg_agrees_with_pow():
    """AVX512-IFMA 8-lane restatement (oracle/ifma_oracle.c, CPU baseline B3) vs CPython pow: every
    modulus width of the BASELINE configs, ragged batch sizes (the 8-lane tail), exponent 0/1, base 0."""
    if c_oracle.ifma_lib() is None:
        pytest.skip("no avx512ifma on this host (or compiler without -mavx512ifma)")
    rng = random.Random(17)
    for bits, ebits, cnt in ((1024, 512, 9), (2048, 1024, 17), (3072, 1536, 8), (4096, 1024, 11), (4096, 2048, 3),
                             (6144, 1536, 5), (4096, 32, 7), (1000, 333, 10), (4000, 77, 1)):
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        W, E = (bits + 63) // 64, (ebits + 63) // 64
        base = [rng.randrange(mod) for _ in range(cnt)]
        exp = [rng.getrandbits(ebits) for _ in range(cnt)]
        exp[0] = 0
        base[-1] = 0 if cnt > 2 else base[-1]
        if cnt > 3:
            exp[1], base[2] = 1, mod - 1
        got = limbs_to_ints(c_oracle.ifma_modexp_batch(ints_to_limbs(base, W), ints_to_limbs(exp, E),
                                                       ints_to_limbs([mod], W)[0]))
        assert got == [pow(b, e, mod) for b, e in zip(base, exp)], (bits, ebits)


def test_ifma_fixed_base_leg_agrees_with_pow_and_square_and_multiply():
    """oracle/ifma_oracle.c: orc_ifma_fb_* (round 4, cpu_baseline.legs.ifma_fixed_base): base^e through a per-base table of
    base^(d * 2^(w i)) -- the CPU counterpart of the GPU's DJN obfuscator -- against CPython pow and against the
    square-and-multiply leg, edge exponents and the 8-lane tail included; DJN encrypt of the benchmark key through it
    equals the scalar port's."""
    if c_oracle.ifma_lib() is None or not hasattr(c_oracle.ifma_lib(), "orc_ifma_fb_build"):
        pytest.skip("no avx512ifma on this host")
    rng = random.Random(77)
    for bits, ebits, w, cnt in ((4096, 1024, 8, 21), (4096, 1024, 5, 9), (2048, 512, 10, 16), (1000, 333, 7, 3), (6144, 1536, 8, 5)):
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        base = rng.randrange(mod)
        exp = [rng.getrandbits(ebits) for _ in range(cnt)]
        exp[0], exp[1], exp[-1] = 0, 1, (1 << ebits) - 1
        W, E = (bits + 63) // 64, (ebits + 63) // 64
        fb = c_oracle.IfmaFixedBase(ints_to_limbs([base], W)[0], ints_to_limbs([mod], W)[0], ebits, w)
        got = limbs_to_ints(fb(None, ints_to_limbs(exp, E)))
        fb.close()
        assert got == [pow(base, e, mod) for e in exp], (bits, ebits, w)
        ref = limbs_to_ints(c_oracle.ifma_modexp_batch(ints_to_limbs([base] * cnt, W), ints_to_limbs(exp, E),
                                                       ints_to_limbs([mod], W)[0]))
        assert got == ref
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
    n = p * q
    m = [rng.randrange(n) for _ in range(19)]
    r = [rng.getrandbits(1024) for _ in range(19)]
    n_l, hs_l = ints_to_limbs([n], 32)[0], ints_to_limbs([hs], 64)[0]
    fb = c_oracle.IfmaFixedBase(hs_l, ints_to_limbs([n * n], 64)[0], 1024, 8)
    via_fb = c_oracle.paillier_encrypt_with(fb, n_l, hs_l, ints_to_limbs(m, 32), ints_to_limbs(r, 16))
    fb.close()
    assert np.array_equal(via_fb, c_oracle.paillier_encrypt(n_l, hs_l, ints_to_limbs(m, 32), ints_to_limbs(r, 16)))


def test_ifma_leg_on_iso_kat():
    """r^n mod n^2 of the reference's ISO/IEC 18033-6 vector (test/test_cryptography.cpp:99-241)."""
    if c_oracle.ifma_lib() is None:
        pytest.skip("no avx512ifma on this host")
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q = int(k["p"], 16), int(k["q"], 16)
    n = p * q
    r = [int(k["r0"], 16), int(k["r1"], 16)]
    got = limbs_to_ints(c_oracle.ifma_modexp_batch(ints_to_limbs(r, 64), ints_to_limbs([n, n], 32),
                                                   ints_to_limbs([n * n], 64)[0]))
    assert got == [pow(x, n, n * n) for x in r]


@pytest.mark.parametrize("which", ["ifma", "openssl"])
def test_split_flows_match_scalar_port_and_kat(which):
    """encrypt / CRT decrypt with the modexps done by the IFMA or OpenSSL leg and the host glue by
    oracle/modexp_oracle.c: bit-identical to the scalar port and to the reference's ISO KAT."""
    backend = {"ifma": (c_oracle.ifma_lib, c_oracle.ifma_modexp_batch),
               "openssl": (c_oracle.openssl_lib, c_oracle.openssl_modexp_batch)}[which]
    if backend[0]() is None:
        pytest.skip(which + " leg not available on this host")
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q = sorted((int(k["p"], 16), int(k["q"], 16)))
    n = p * q
    sk = orc.PrivateKey(n, p, q)
    rng = random.Random(23)
    m = [int(k["m0"], 16), int(k["m1"], 16)] + [rng.randrange(n) for _ in range(9)]
    r = [int(k["r0"], 16), int(k["r1"], 16)] + [rng.randrange(1, n) for _ in range(9)]
    n_l, m_l, r_l = ints_to_limbs([n], 32)[0], ints_to_limbs(m, 32), ints_to_limbs(r, 32)
    c = c_oracle.paillier_encrypt_with(backend[1], n_l, None, m_l, r_l)
    assert np.array_equal(c, c_oracle.paillier_encrypt(n_l, None, m_l, r_l))
    assert limbs_to_ints(c[:2]) == [int(k["c1"], 16), int(k["c2"], 16)]
    hs = int(k["bench_hs"], 16)
    hs_l, rs_l = ints_to_limbs([hs], 64)[0], ints_to_limbs([rng.getrandbits(1024) for _ in m], 16)
    cd = c_oracle.paillier_encrypt_with(backend[1], n_l, hs_l, m_l, rs_l)
    assert np.array_equal(cd, c_oracle.paillier_encrypt(n_l, hs_l, m_l, rs_l))
    args = [ints_to_limbs([v], 16)[0] for v in (sk.p, sk.q, sk.hp, sk.hq, sk.pinv)]
    for ct in (c, cd):
        assert limbs_to_ints(c_oracle.paillier_decrypt_crt_with(backend[1], *args, ct)) == m
