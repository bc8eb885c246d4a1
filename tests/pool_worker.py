"""Scenario runner for tests/test_gpu_pool.py: runs in its OWN process (a pool is a process-wide choice), checks
everything against the oracle (the checker) and prints one JSON line.  usage: pool_worker.py <scenario> <n_devices>"""
import ctypes
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def is_prime(v, rng):
    if v < 2 or v % 2 == 0:
        return v == 2
    for sp in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if v % sp == 0:
            return v == sp
    d, s = v - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(24):
        a = rng.randrange(2, v - 1)
        x = pow(a, d, v)
        if x in (1, v - 1):
            continue
        for _ in range(s - 1):
            x = x * x % v
            if x == v - 1:
                break
        else:
            return False
    return True


def gen_unequal_primes(bits):
    """primes p < q with p^2 of bits-1 bits, q^2 and p*q of `bits` bits"""
    rng = random.Random(bits)
    half = bits // 2
    lo_p, hi_p = int(2 ** 0.40 * (1 << 20)) << (half - 21), int(2 ** 0.45 * (1 << 20)) << (half - 21)
    lo_q, hi_q = int(2 ** 0.80 * (1 << 20)) << (half - 21), int(2 ** 0.90 * (1 << 20)) << (half - 21)

    def pick(lo, hi):
        while True:
            v = rng.randrange(lo, hi) | 1
            if is_prime(v, rng):
                return v
    return pick(lo_p, hi_p), pick(lo_q, hi_q)


def main():
    scenario, ndev = sys.argv[1], int(sys.argv[2])
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    import pailliercryptolib_amd as pa
    L = _capi.lib()
    _capi.check(L.pgpu_init_all(ndev))
    pa.engine._initialized = True
    res = {"pool": L.pgpu_pool_size(), "transport": L.pgpu_pool_transport().decode()}
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
    n = p * q
    nsq = n * n
    nw = 32
    rng = random.Random(77)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    if scenario == "host_api":
        # every host-pointer entry point with ragged shard sizes; order must be preserved
        ok = {}
        for count in (1, 5, 61, 130):
            m = [rng.randrange(n) for _ in range(count)]
            r = [rng.getrandbits(1024) for _ in range(count)]
            pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
            opk = orc.PublicKey(n, 2048)
            opk.set_djn(hs)
            ct = pk.encrypt(m, r)
            ok[f"enc{count}"] = ct == opk.encrypt(m, r)
            ok[f"dec{count}"] = sk.decrypt(ct) == m
            ok[f"mul{count}"] = pa.mod_mul(ct, ct[::-1], nsq) == [a * b % nsq for a, b in zip(ct, ct[::-1])]
            e = [rng.getrandbits(64) for _ in range(count)]
            ok[f"exp{count}"] = pa.mod_exp(ct, e, nsq) == [pow(a, b, nsq) for a, b in zip(ct, e)]
            pk2 = pa.PublicKey(n, 2048)            # non-DJN: r^n
            rr = [rng.randrange(1, n) for _ in range(count)]
            ok[f"encn{count}"] = pk2.encrypt(m, rr) == orc.PublicKey(n, 2048).encrypt(m, rr)
        res["ok"] = ok
    elif scenario == "batch_chain":
        count = 203
        m1 = [rng.randrange(n) for _ in range(count)]
        m2 = [rng.randrange(n) for _ in range(count)]
        r1 = [rng.getrandbits(1024) for _ in range(count)]
        r2 = [rng.getrandbits(1024) for _ in range(count)]
        e = [rng.getrandbits(40) for _ in range(count)]
        pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)

        def up(vals, words):
            h = ctypes.c_void_p()
            a = ints_to_limbs(vals, words)
            _capi.check(L.pgpu_batch_upload(ptr(a), len(vals), words, words, ctypes.byref(h)))
            return h

        def down(h):
            out = np.empty((L.pgpu_batch_count(h), L.pgpu_batch_words(h)), dtype=np.uint64)
            _capi.check(L.pgpu_batch_download(h, ptr(out)))
            return limbs_to_ints(out)

        def op(fn, *a):
            h = ctypes.c_void_p()
            _capi.check(fn(*a, ctypes.byref(h)))
            return h
        bm1, bm2, br1, br2, be = up(m1, nw), up(m2, nw), up(r1, 16), up(r2, 16), up(e, 1)
        c1 = op(L.pgpu_batch_encrypt, pk._h, bm1, br1, 1024)
        c2 = op(L.pgpu_batch_encrypt, pk._h, bm2, br2, 1024)
        opk = orc.PublicKey(n, 2048)
        opk.set_djn(hs)
        oc1, oc2 = opk.encrypt(m1, r1), opk.encrypt(m2, r2)
        ok = {"mont": bool(L.pgpu_batch_is_montgomery(c1)), "c1": down(c1) == oc1, "c2": down(c2) == oc2}
        s = op(L.pgpu_batch_ct_add, pk._h, c1, c2)
        osum = [a * b % nsq for a, b in zip(oc1, oc2)]
        ok["add"] = down(s) == osum
        t = op(L.pgpu_batch_ct_mul, pk._h, s, be, 40)
        omul = [pow(a, b, nsq) for a, b in zip(osum, e)]
        ok["mul"] = down(t) == omul
        u = op(L.pgpu_batch_ct_add_plain, pk._h, t, bm2)
        oadd = [a * ((1 + n * b) % nsq) % nsq for a, b in zip(omul, m2)]
        ok["addpt"] = down(u) == oadd
        d = op(L.pgpu_batch_decrypt_crt, sk._h, u)
        want = [((a + b) * x + b) % n for a, b, x in zip(m1, m2, e)]
        ok["dec"] = down(d) == want
        # mixed forms: a plain (uploaded) ciphertext batch joins the chain; a one-element operand broadcasts
        pc2 = up(oc2, 2 * nw)
        ok["add_mixed"] = down(op(L.pgpu_batch_ct_add, pk._h, c1, pc2)) == osum
        one = up([oc2[0]], 2 * nw)
        ok["add_bcast"] = down(op(L.pgpu_batch_ct_add, pk._h, c1, one)) == [a * oc2[0] % nsq for a in oc1]
        e1 = up([e[0]], 1)
        ok["mul_bcast"] = down(op(L.pgpu_batch_ct_mul, pk._h, pc2, e1, 40)) == [pow(a, e[0], nsq) for a in oc2]
        ok["dec_plain"] = down(op(L.pgpu_batch_decrypt_crt, sk._h, pc2)) == m2
        res["ok"] = ok
    elif scenario == "no_terminate":
        # a script that forgets pgpu_shutdown: the worker lanes must not keep the process from exiting
        m = [rng.randrange(n) for _ in range(20)]
        pk = pa.PublicKey(n, 2048, hs=hs)
        res["ok"] = {"enc": len(pk.encrypt(m, [rng.getrandbits(1024) for _ in m])) == 20}
    elif scenario == "unequal_key":
        # p^2 one bit shorter than q^2, straddling the unit-quotient-digit headroom of the geometry: both contexts of
        # a decrypt launch must agree on the loop form (odd-parity waves read a null nhat otherwise)
        ok = {}
        for bits in (1008, 1588):
            p2, q2 = gen_unequal_primes(bits)
            n2 = p2 * q2
            assert (p2 * p2).bit_length() == bits - 1 and (q2 * q2).bit_length() == bits and n2.bit_length() == bits
            m = [rng.randrange(n2) for _ in range(70)]
            r = [rng.randrange(1, n2) for _ in range(70)]
            ct = pa.PublicKey(n2, bits).encrypt(m, r)
            ok[f"enc{bits}"] = ct == orc.PublicKey(n2, bits).encrypt(m, r)
            ok[f"dec{bits}"] = pa.PrivateKey(p2, q2).decrypt(ct) == m
        res["ok"] = ok
    elif scenario == "corrupt_replica":
        # A collective that "succeeds" with wrong bytes on one rank: the debug hook flips a byte of
        # the NEXT replicated key image on the last pool entry right after the broadcast.  The read-back check must see
        # it, rewrite the copy and count it -- and every shard must still decrypt correctly.
        ver0, rep0 = ctypes.c_uint64(), ctypes.c_uint64()
        L.pgpu_replication_stats(ctypes.byref(ver0), ctypes.byref(rep0))
        _capi.check(L.pgpu_debug_corrupt_next_replica(ndev - 1))
        sk = pa.PrivateKey(p, q)                 # its first image (mod p^2 context) is the one that gets hit
        _capi.check(L.pgpu_debug_corrupt_next_replica(ndev - 1))
        pk = pa.PublicKey(n, 2048, hs=hs)
        ver, rep = ctypes.c_uint64(), ctypes.c_uint64()
        L.pgpu_replication_stats(ctypes.byref(ver), ctypes.byref(rep))
        count = 40 * ndev
        m = [rng.randrange(n) for _ in range(count)]
        r = [rng.getrandbits(1024) for _ in range(count)]
        opk = orc.PublicKey(n, 2048)
        opk.set_djn(hs)
        ct = pk.encrypt(m, r)
        res["ok"] = {"detected": rep.value - rep0.value == 2, "verified": ver.value - ver0.value >= 2 * ndev,
                     "enc": ct == opk.encrypt(m, r), "dec": sk.decrypt(ct) == m}
        res["repaired"] = rep.value - rep0.value
    elif scenario == "reinit":
        # objects of a pool that was shut down are refused (stale Replicated images were indexed blindly);
        # the per-modulus caches of the key-less seam do not survive the pool either
        ok = {}
        pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
        m = [rng.randrange(n) for _ in range(9)]
        r = [rng.getrandbits(1024) for _ in range(9)]
        ct = pk.encrypt(m, r)
        ok["before"] = sk.decrypt(ct) == m
        psq = p * p
        ctp = [c % psq for c in ct]                       # (the seam takes bases below the modulus)
        ok["seam_before"] = pa.mod_exp(ctp, [p - 1] * 9, psq) == [pow(c, p - 1, psq) for c in ctp]
        L.pgpu_shutdown()
        _capi.check(L.pgpu_init_all(ndev + 1))          # a DIFFERENT device set
        out = np.zeros((9, 64), dtype=np.uint64)
        a_m, a_r = ints_to_limbs(m, 32), ints_to_limbs(r, 16)
        rc = L.pgpu_paillier_encrypt(pk._h, ptr(a_m), 32, 32, ptr(a_r), 16, 16, 1024, ptr(out), 9)
        ok["stale_key_refused"] = rc != 0 and b"shut down" in L.pgpu_last_error()
        pk2, sk2 = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
        ct2 = pk2.encrypt(m, r)
        ok["after"] = ct2 == ct and sk2.decrypt(ct2) == m
        ok["seam_after"] = pa.mod_exp(ctp, [p - 1] * 9, psq) == [pow(c, p - 1, psq) for c in ctp]   # same modulus, new pool
        res["ok"] = ok
        res["pool"] = L.pgpu_pool_size()
    elif scenario == "fb_budget":
        # 64 DJN keys on one GPU under a 64 MiB fixed-base budget: every key's table is built,
        # used and evicted in turn; live table bytes never exceed the budget; ciphertexts are what the oracle says
        budget = 64 << 20
        _capi.check(L.pgpu_set_fixed_base_budget(budget, budget))
        live, ev = ctypes.c_size_t(), ctypes.c_uint64()
        ok, worst = {}, 0
        keys = []
        for i in range(64):
            hs_i = pow(hs, 2 * i + 3, nsq)                # 64 distinct valid hs values (powers of an n-th residue)
            keys.append((pa.PublicKey(n, 2048, hs=hs_i), hs_i))
        m = [rng.randrange(n) for _ in range(6)]
        r = [rng.getrandbits(1024) for _ in range(6)]
        good = True
        for rnd in range(2):
            for pk_i, hs_i in keys:
                opk = orc.PublicKey(n, 2048)
                opk.set_djn(hs_i)
                good = good and pk_i.encrypt(m, r) == opk.encrypt(m, r)
                for d in range(ndev):
                    L.pgpu_fixed_base_stats(d, ctypes.byref(live), ctypes.byref(ev))
                    worst = max(worst, live.value)
        ok["ciphertexts"] = good
        ok["within_budget"] = worst <= budget
        ok["evicted"] = ev.value > 0
        res["ok"] = ok
        res["worst_live_bytes"] = worst
        res["evictions"] = ev.value
    print(json.dumps(res), flush=True)
    if scenario != "no_terminate":
        pa.terminate()


if __name__ == "__main__":
    main()
