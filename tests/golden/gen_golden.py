"""Generates the committed golden fixtures.  Run in the BUILD container only.

  iso_kat.json      -- the constants of the reference's only fixed-vector test,
                       CryptoTest.ISO_IEC_18033_6_ComplianceTest (test/test_cryptography.cpp:99-241),
                       parsed out of the reference test source (data, not code), plus the
                       benchmark key's HS_BN (benchmark/bench_cryptography.cpp:48-63).
  seeded_vectors.json -- seeded batches for other key sizes / paths, produced by the oracle
                       (oracle/paillier_oracle.py, CPython pow) -- "parity unpinned" by the
                       reference itself, see oracle header.
"""
import json
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import paillier_oracle as orc  # noqa: E402

REF = "/root/reference"


def grab(src, name):
    m = re.search(r"BigNumber\s+" + name + r"\s*=\s*((?:\s*\"[0-9a-fx]+\"\s*)+);", src)
    if not m:
        m = re.search(r"const BigNumber\s+" + name + r"\s*=\s*((?:\s*\"[0-9a-fx]+\"\s*)+);", src)
    return "".join(re.findall(r"\"([0-9a-fx]+)\"", m.group(1)))


def is_probable_prime(n, rng, rounds=24):
    if n < 2:
        return False
    for sp in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % sp == 0:
            return n == sp
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(rounds):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_prime(bits, rng):
    """DJN-style prime: top two bits set, p = 3 mod 4 (keygen.cpp:73-90)."""
    while True:
        p = rng.getrandbits(bits) | (3 << (bits - 2)) | 3
        if is_probable_prime(p, rng):
            return p


def gen_key(bits, seed):
    from math import gcd
    rng = random.Random(seed)
    while True:
        p, q = gen_prime(bits // 2, rng), gen_prime(bits // 2, rng)
        if p != q and gcd(p - 1, q - 1) == 2 and (p * q).bit_length() == bits:
            return p, q


def main():
    t = open(os.path.join(REF, "test/test_cryptography.cpp")).read()
    b = open(os.path.join(REF, "benchmark/bench_cryptography.cpp")).read()
    kat = {k: grab(t, k) for k in ("p", "q", "c1", "c2", "c1c2", "m1m2", "r0", "r1")}
    kat["m0"] = "0x414243444546474849404a4b4c4d4e4f"   # test_cryptography.cpp:200
    kat["m1"] = "0x20202020202020202020202020202020"   # test_cryptography.cpp:203
    kat["num_values"] = 21                               # test_cryptography.cpp:102
    kat["bench_hs"] = grab(b, "HS_BN")
    kat["bench_r"] = grab(b, "R_BN")
    kat["source"] = ("reference test/test_cryptography.cpp:104-203 (ISO/IEC 18033-6 KAT), "
                     "benchmark/bench_cryptography.cpp:24-63")
    json.dump(kat, open(os.path.join(HERE, "iso_kat.json"), "w"), indent=1)

    out = {"generator": "tests/golden/gen_golden.py + oracle/paillier_oracle.py (CPython pow)", "cases": []}
    for bits, seed, count in ((1024, 1024, 8), (2048, 2048, 6), (3072, 3072, 4)):
        p, q = gen_key(bits, seed)
        n = p * q
        rng = random.Random(seed + 1)
        for djn in (True, False):
            pk = orc.PublicKey(n, bits)
            if djn:
                x = rng.randrange(2, n)
                pk.set_djn(pk.hs_from_x(x))
            sk = orc.PrivateKey(n, p, q)
            m = [rng.randrange(n) for _ in range(count)]
            m[0], m[1] = 0, n - 1
            r = [rng.getrandbits(bits // 2) if djn else rng.randrange(1, n) for _ in range(count)]
            c = pk.encrypt(m, r)
            assert sk.decrypt(c) == m and sk.decrypt(c, crt=False) == m
            e32 = [rng.getrandbits(32) for _ in range(count)]
            e32[0] = 0
            out["cases"].append({
                "bits": bits, "djn": djn, "p": hex(p), "q": hex(q), "hs": hex(pk.hs),
                "m": [hex(v) for v in m], "r": [hex(v) for v in r], "c": [hex(v) for v in c],
                "add": [hex(v) for v in orc.ct_add(c, c[::-1], n * n)],
                "mul_exp": [hex(v) for v in e32],
                "mul": [hex(v) for v in orc.ct_mul_pt(c, e32, n * n)],
            })
    json.dump(out, open(os.path.join(HERE, "seeded_vectors.json"), "w"), indent=1)
    print("wrote fixtures")


if __name__ == "__main__":
    main()
