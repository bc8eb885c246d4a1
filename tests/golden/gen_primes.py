"""Writes tests/golden/primes_4096.json: two seeded 2048-bit primes (a 4096-bit Paillier key) for the parity tests
of the widest compiled key class, and tests/golden/primes_uneven.json: a 1013-bit and a 1024-bit prime (a
ciphertext then enters the split-form kernel in five 960-bit chunks instead of four 1024-bit ones).  Not
reference vectors: the expected values come from the oracle at test time.
usage: python tests/golden/gen_primes.py"""
import json
import os
import random

SMALL = [p for p in range(3, 2000, 2) if all(p % d for d in range(3, int(p ** 0.5) + 1, 2))]


def is_prime(n, rng):
    if any(n % p == 0 for p in SMALL):
        return False
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(24):
        x = pow(rng.randrange(2, n - 1), d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def prime(bits, rng, top2=True):
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | ((1 << (bits - 2)) if top2 else 0) | 1
        if is_prime(c, rng):
            return c


if __name__ == "__main__":
    rng = random.Random(4096)
    p, q = prime(2048, rng), prime(2048, rng)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "primes_4096.json")
    json.dump({"bits": 4096, "p": hex(min(p, q)), "q": hex(max(p, q))}, open(out, "w"), indent=1)
    print(out)
    rng = random.Random(1013)
    p, q = prime(1013, rng, top2=False), prime(1024, rng, top2=False)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "primes_uneven.json")
    json.dump({"bits": (p * q).bit_length(), "p": hex(p), "q": hex(q)}, open(out, "w"), indent=1)
    print(out)
