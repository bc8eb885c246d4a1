"""Differential soak from SEVERAL host threads at once (round 6: the two multi-thread bugs of profiles/r06_thread_race.txt were
found by a benchmark, not by the single-threaded soaks).  T Python threads (ctypes releases the GIL inside the library) share a
few keys -- as the reference's OpenMP tests share one (test_cryptography.cpp:45-57) -- and run, on batches of random small and
middle sizes: encrypt (sample against the oracle), CRT decrypt of the resident ciphertexts and from host arrays, CT+CT,
CT x PT, decrypt of the result.  usage: python tools/fuzz_threads.py [seconds] [seed] [threads]"""
import ctypes, os, random, sys, threading, time
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
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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


rng0 = random.Random(seed)
keys = []
for pbits, djn in ((1024, True), (1024, False), (512, True), (1536, True)):
    while True:
        p, q = gen_primes.prime(pbits, rng0, top2=True), gen_primes.prime(pbits, rng0, top2=True)
        if p != q:
            break
    n = p * q
    bits = n.bit_length()
    opk = orc.PublicKey(n, bits)
    hs = None
    if djn:
        x = rng0.randrange(2, n)
        hs = pow((-x * x) % n, n, n * n)
        opk.set_djn(hs)
    keys.append(dict(n=n, bits=bits, djn=djn, opk=opk, pk=pa.PublicKey(n, bits, hs=hs), sk=pa.PrivateKey(p, q)))

errors, cases = [], [0] * T
deadline = time.time() + budget
start = threading.Barrier(T)


def worker(t):
    rng = random.Random(seed * 1000 + t)
    start.wait()
    try:
        while time.time() < deadline and not errors:
            k = rng.choice(keys)
            n, bits, djn, pk, sk = k["n"], k["bits"], k["djn"], k["pk"], k["sk"]
            nw = (bits + 63) // 64
            count = rng.choice([1, 3, 16, 64, 200, 512, 700, 1024, 1500, 2049, rng.randrange(1, 4200)])
            rb = bits // 2 if djn else bits
            m = [rng.randrange(n) for _ in range(count)]
            r = [rng.getrandbits(rb) if djn else rng.randrange(1, n) for _ in range(count)]
            tag = (t, cases[t], bits, djn, count, seed)
            hm, hr, c = up(m, nw), up(r, (rb + 63) // 64), ctypes.c_void_p()
            _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, rb, ctypes.byref(c)))
            ct = down(c)
            for i in sorted(set([0, count - 1, count // 2])):
                assert ct[i] == k["opk"].encrypt([m[i]], [r[i]])[0], ("encrypt",) + tag + (i,)
            d = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))
            got = down(d)
            assert got == m, ("roundtrip",) + tag + ([i for i in range(count) if got[i] != m[i]][:8],)
            if cases[t] % 3 == 1:
                got = pk.encrypt(m, r)            # the host-array entry point (pgpu_paillier_encrypt) against the resident path
                assert got == ct, ("encrypt from host",) + tag + ([i for i in range(count) if got[i] != ct[i]][:8],)
            if cases[t] % 3 == 0:
                got = sk.decrypt(ct)
                assert got == m, ("decrypt from host",) + tag + ([i for i in range(count) if got[i] != m[i]][:8],)
            s, e, pr = ctypes.c_void_p(), up([rng.getrandbits(40) for _ in range(count)], 1), ctypes.c_void_p()
            _capi.check(L.pgpu_batch_ct_add(pk._h, c, c, ctypes.byref(s)))
            _capi.check(L.pgpu_batch_ct_mul(pk._h, s, e, 40, ctypes.byref(pr)))
            d2 = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_decrypt_crt(sk._h, pr, ctypes.byref(d2)))
            ev, got = down(e), down(d2)
            want = [(2 * a * b) % n for a, b in zip(m, ev)]
            assert got == want, ("ops",) + tag + ([i for i in range(count) if got[i] != want[i]][:8],)
            for h in (hm, hr, c, d, s, e, pr, d2):
                L.pgpu_batch_destroy(h)
            cases[t] += 1
    except BaseException as ex:   # noqa: a failed comparison in any thread ends the soak
        errors.append(repr(ex)[:600])


threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
for th in threads:
    th.start()
for th in threads:
    th.join()
if errors:
    print("THREAD FUZZ FAILED:", errors[0])
    sys.exit(1)
print(f"thread fuzz ok: {T} threads, {sum(cases)} cases ({cases}), seed {seed}, {time.time() - t0:.0f} s")
keys.clear()
pa.terminate()
