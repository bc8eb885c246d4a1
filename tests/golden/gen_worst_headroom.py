"""Generates tests/golden/primes_worst_headroom.json: two 1536-bit primes p, q == 1 (mod 2^28) just below 2^1536.
For such a prime the multiplier of the split form, k = -p^-1 mod 2^28, is 2^28 - 1, so P = p * k sits just below
2^(1536 + 28): the LARGEST P a 3072-bit key can have, and R = 2^(28 * 56) = 16 * 2^(1536 + 28) the tightest headroom
the one-lane product-scanning decrypt kernel (csrc/hensel_ps.hpp: R >= 16 P) ever sees.  Data only (inputs of a GPU
parity test against the oracle); deterministic.  usage: python tests/golden/gen_worst_headroom.py"""
import json
import os
import random

import gen_primes

rng = random.Random(56)
out = []
j = 1
while len(out) < 2:
    c = (1 << 1536) - (j << 28) + 1
    j += 1
    if gen_primes.is_prime(c, rng):
        assert (-pow(c, -1, 1 << 28)) % (1 << 28) == (1 << 28) - 1
        out.append(c)
json.dump({"bits": 3072, "p": hex(out[0])[2:], "q": hex(out[1])[2:],
           "note": "p, q == 1 mod 2^28 just below 2^1536: k = 2^28 - 1, P = p k just below 2^1564 (gen_worst_headroom.py)"},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "primes_worst_headroom.json"), "w"), indent=1)
print(j, [x.bit_length() for x in out])
