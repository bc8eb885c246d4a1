"""Randomised differential soak of the Paillier entry points (keys of random sizes, DJN or not, random batch sizes)
against the oracle: encrypt on a sample, full round trip, CT+CT, CT x PT on resident batches (diagnostics; the
contract tests are in tests/).  usage: python tools/fuzz_paillier.py [seconds] [seed]"""
import ctypes, os, random, sys, time
sys.path.insert(0, ".")
sys.path.insert(0, os.path.join("tests", "golden"))
import numpy as np
import gen_primes
import pailliercryptolib_amd as pa
from oracle import paillier_oracle as orc
from pailliercryptolib_amd import _capi
from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
pa.initialize()
L = _capi.lib()


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def up(vals, words):
    h = ctypes.c_void_p()
    a = ints_to_limbs(vals, words)
    _capi.check(L.pgpu_batch_upload(ptr(a), len(vals), words, words, ctypes.byref(h)))
    return h


def down(h):
    out = np.empty((L.pgpu_batch_count(h), L.pgpu_batch_words(h)), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(h, ptr(out)))
    return limbs_to_ints(out)


t0, cases = time.time(), 0
while time.time() - t0 < budget:
    pbits = rng.choice([256, 320, 512, 640, 768, 1024, 1024, 1024, 1280, 1536])
    p = gen_primes.prime(pbits, rng, top2=True)
    q = gen_primes.prime(pbits, rng, top2=True)
    if p == q:
        continue
    n = p * q
    bits = n.bit_length()
    nw = (bits + 63) // 64
    djn = rng.random() < 0.6
    opk = orc.PublicKey(n, bits)
    hs = None
    if djn:
        x = rng.randrange(2, n)
        hs = pow((-x * x) % n, n, n * n)                    # pub_key.cpp:31-38
        opk.set_djn(hs)
    pk = pa.PublicKey(n, bits, hs=hs)
    sk = pa.PrivateKey(p, q)
    count = rng.choice([1, 3, 17, 100, 2049, 4097, rng.randrange(1, 9000)])
    rb = bits // 2 if djn else bits
    m = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(rb) if djn else rng.randrange(1, n) for _ in range(count)]
    idx = sorted(set([0, count - 1, count // 2]))
    hm, hr, c = up(m, nw), up(r, (rb + 63) // 64), ctypes.c_void_p()
    _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, rb, ctypes.byref(c)))
    ct = down(c)
    for i in idx:
        assert ct[i] == opk.encrypt([m[i]], [r[i]])[0], ("encrypt", bits, djn, count, i, seed, cases)
    d = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))
    assert down(d) == m, ("roundtrip", bits, djn, count, seed, cases)
    assert sk.decrypt(ct) == m, ("decrypt from host", bits, djn, count, seed, cases)
    s, e, t = ctypes.c_void_p(), up([rng.getrandbits(40) for _ in range(count)], 1), ctypes.c_void_p()
    _capi.check(L.pgpu_batch_ct_add(pk._h, c, c, ctypes.byref(s)))
    _capi.check(L.pgpu_batch_ct_mul(pk._h, s, e, 40, ctypes.byref(t)))
    d2 = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_decrypt_crt(sk._h, t, ctypes.byref(d2)))
    ev = down(e)
    assert down(d2) == [(2 * a * b) % n for a, b in zip(m, ev)], ("ops", bits, djn, count, seed, cases)
    for h in (hm, hr, c, d, s, e, t, d2):
        L.pgpu_batch_destroy(h)
    del pk, sk
    cases += 1
print(f"paillier fuzz ok: {cases} cases, seed {seed}, {time.time() - t0:.0f} s")
pa.terminate()
