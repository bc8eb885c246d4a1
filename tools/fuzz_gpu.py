"""Randomised differential soak of the C-ABI against OpenSSL BN_mod_exp_mont / CPython (diagnostics; the
contract tests are in tests/).  usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import random, sys, time
import numpy as np
sys.path.insert(0, ".")
import pailliercryptolib_amd as pa
from oracle import c_oracle
from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
pa.initialize()
have_ossl = c_oracle.openssl_lib() is not None
t0, cases, elems = time.time(), 0, 0
while time.time() - t0 < budget:
    bits = rng.choice([rng.randrange(65, 8192), rng.choice([512, 1024, 2048, 3072, 4096, 6144, 8192])])
    style = rng.randrange(6)
    if style >= 4:                                                      # perfect squares: the split form of the seam
        root = rng.getrandbits(bits // 2) | (1 << (bits // 2 - 1)) | 1
        mod = root * root
        bits = mod.bit_length()
    elif style == 0:
        mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    elif style == 1:
        mod = (1 << bits) - rng.randrange(1, 1 << 20, 2)            # all-ones limbs
    elif style == 2:
        mod = (1 << (bits - 1)) + rng.randrange(1, 1 << 20, 2)        # sparse
    else:
        mod = (rng.getrandbits(bits) | (1 << (bits - 1)) | 1) & ~((1 << (bits // 2)) - (1 << 3)) | 1
    W = (bits + 63) // 64
    count = rng.choice([1, 2, 7, 8, 9, 16, 17, 33, 64, 100, 257, rng.randrange(1, 600)])
    ebits = rng.choice([1, 5, 32, 64, 100, 300, 1024, min(2048, bits), rng.randrange(1, 1500)])
    if rng.random() < 0.08 and bits <= 4096:                            # batches that take the throughput kernel forms
        count = rng.choice([4097, 5000, 8193, 9000, 17000])
        ebits = rng.choice([1, 17, 40])
    shared = rng.random() < 0.4
    base = [rng.choice([mod - 1, 0, 1, rng.randrange(mod), (1 << (bits - 1)) - 1]) if rng.random() < 0.1
            else rng.randrange(mod) for _ in range(count)]
    exps = [rng.getrandbits(ebits)] if shared else [rng.getrandbits(ebits) for _ in range(count)]
    got = pa.engine.mod_exp(base, exps, mod)
    E = max(1, (ebits + 63) // 64)
    if have_ossl and bits % 64 == 0 and count > 8:
        e_full = exps * count if shared else exps
        want = limbs_to_ints(c_oracle.openssl_modexp_batch(ints_to_limbs(base, W), ints_to_limbs(e_full, E),
                                                           ints_to_limbs([mod], W)[0]))
    else:
        want = [pow(b, exps[0] if shared else e, mod) for b, e in zip(base, exps * count if shared else exps)]
    if got != want:
        bad = [i for i in range(count) if got[i] != want[i]][:3]
        print("MISMATCH", dict(bits=bits, style=style, count=count, ebits=ebits, shared=shared, idx=bad, seed=seed, case=cases))
        print("mod", hex(mod)); print("base", hex(base[bad[0]])); print("exp", hex(exps[0] if shared else exps[bad[0]]))
        sys.exit(1)
    if rng.random() < 0.3:                                          # a product on the same modulus
        other = [rng.randrange(mod) for _ in range(count)]
        assert pa.engine.mod_mul(base, other, mod) == [(x * y) % mod for x, y in zip(base, other)], ("modmul", bits, count)
    cases += 1
    elems += count
print(f"fuzz ok: {cases} cases, {elems} exponentiations, seed {seed}, {time.time() - t0:.0f} s")
