"""Resident ciphertext batches as PAIR ROWS (csrc/kargs.hpp, round 3): keys with a split form keep c*R as the pair
(a, b), c*R == a - (n*k)*b mod n^2, in HBM -- the register image of the split-form kernels.  Every operation on
resident batches (encrypt DJN / r^n, CT+CT, CT+PT, CT x PT, CRT decrypt, download, mixing with uploaded plain
ciphertexts, one-element broadcast operands) is held bit-identical to the oracle for the 1024-, 2048- and 3072-bit key
classes, at small and medium batch sizes (the large-batch forms are covered element by element against the C oracle in
test_gpu_sha256_fullsize.py), with edge plaintexts and exponents."""
import ctypes
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Res:
    def __init__(self):
        from pailliercryptolib_amd import _capi
        from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
        self.L, self.check, self.i2l, self.l2i = _capi.lib(), _capi.check, ints_to_limbs, limbs_to_ints
        self.live = []

    def up(self, vals, words):
        h = ctypes.c_void_p()
        a = self.i2l(vals, words)
        self.check(self.L.pgpu_batch_upload(a.ctypes.data_as(ctypes.c_void_p), len(vals), words, words, ctypes.byref(h)))
        self.live.append(h)
        return h

    def down(self, h):
        out = np.empty((self.L.pgpu_batch_count(h), self.L.pgpu_batch_words(h)), dtype=np.uint64)
        self.check(self.L.pgpu_batch_download(h, out.ctypes.data_as(ctypes.c_void_p)))
        return self.l2i(out)

    def op(self, fn, *a):
        h = ctypes.c_void_p()
        self.check(fn(*a, ctypes.byref(h)))
        self.live.append(h)
        return h

    def close(self):
        for h in self.live:
            self.L.pgpu_batch_destroy(h)
        self.live = []


def key_case(bits, djn):
    if bits == 2048:
        k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
        return int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16) if djn else None
    if bits == 4096:   # (no DJN fixture of this size: hs = (-x^2)^n mod n^2 as pub_key.cpp:40-49 forms it, x seeded)
        k = json.load(open(os.path.join(GOLD, "primes_4096.json")))
        p, q = int(k["p"], 16), int(k["q"], 16)
        n = p * q
        x = random.Random(4096).randrange(2, n)
        return p, q, pow(n * n - x * x % (n * n), n, n * n) if djn else None
    c = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == bits and c["djn"]][0]
    return int(c["p"], 16), int(c["q"], 16), int(c["hs"], 16) if djn else None


@pytest.mark.parametrize("bits,djn,count", [(2048, True, 203), (2048, False, 37), (2048, True, 700), (1024, True, 131),
                                            (1024, False, 19), (3072, True, 70), (3072, False, 9), (2048, True, 1),
                                            (4096, True, 41), (4096, False, 5)])
def test_resident_chain_in_pair_rows(engine, bits, djn, count):
    from oracle import paillier_oracle as orc
    p, q, hs = key_case(bits, djn)
    n = p * q
    nsq = n * n
    nw, pw = bits // 64, bits // 128
    rng = random.Random(bits * 7 + count)
    edge_m = [0, 1, n - 1, n, (1 << (64 * nw)) - 1]           # (rows no wider than n may still hold values >= n)
    m1 = (edge_m + [rng.randrange(n) for _ in range(count)])[:count]
    m2 = [rng.randrange(n) for _ in range(count)]
    if djn:
        r1 = ([0, 1, (1 << (bits // 2)) - 1] + [rng.getrandbits(bits // 2) for _ in range(count)])[:count]
        r2 = [rng.getrandbits(bits // 2) for _ in range(count)]
    else:
        r1 = ([1, n - 1] + [rng.randrange(1, n) for _ in range(count)])[:count]
        r2 = [rng.randrange(1, n) for _ in range(count)]
    e = ([0, 1, 2, (1 << 40) - 1] + [rng.getrandbits(40) for _ in range(count)])[:count]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, bits)
    if djn:
        opk.set_djn(hs)
    R = Res()
    L = R.L
    try:
        rw = pw if djn else nw
        rb = 64 * rw
        bm1, bm2, br1, br2, be = R.up(m1, nw), R.up(m2, nw), R.up(r1, rw), R.up(r2, rw), R.up(e, 1)
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, bm1, br1, rb)
        c2 = R.op(L.pgpu_batch_encrypt, pk._h, bm2, br2, rb)
        oc1, oc2 = opk.encrypt(m1, r1), opk.encrypt(m2, r2)
        assert L.pgpu_batch_is_montgomery(c1) == 1                      # a device-side domain, not plain words
        assert R.down(c1) == oc1 and R.down(c2) == oc2
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
        osum = [a * b % nsq for a, b in zip(oc1, oc2)]
        assert R.down(s) == osum
        t = R.op(L.pgpu_batch_ct_mul, pk._h, s, be, 40)
        omul = [pow(a, b, nsq) for a, b in zip(osum, e)]
        assert R.down(t) == omul
        u = R.op(L.pgpu_batch_ct_add_plain, pk._h, t, bm1)             # plaintexts with the edge values
        oadd = [a * ((1 + n * b) % nsq) % nsq for a, b in zip(omul, m1)]
        assert R.down(u) == oadd
        d = R.op(L.pgpu_batch_decrypt_crt, sk._h, u)
        assert R.down(d) == [((a + b) * x + a) % n for a, b, x in zip(m1, m2, e)]
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c1)) == [v % n for v in m1]
        # an uploaded (plain) ciphertext batch joins; one-element operands broadcast
        pc2 = R.up(oc2, 2 * nw)
        assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c1, pc2)) == osum
        assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, pc2, c1)) == osum
        if count > 1:
            one = R.up([oc2[0]], 2 * nw)
            assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c1, one)) == [a * oc2[0] % nsq for a in oc1]
            e1 = R.up([e[-1]], 1)
            assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c2, e1, 40)) == [pow(a, e[-1], nsq) for a in oc2]
            m0 = R.up([m2[0]], nw)
            assert R.down(R.op(L.pgpu_batch_ct_add_plain, pk._h, c1, m0)) == [a * ((1 + n * m2[0]) % nsq) % nsq for a in oc1]
        assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, pc2, be, 40)) == [pow(a, b, nsq) for a, b in zip(oc2, e)]
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, pc2)) == m2
        # plaintext rows WIDER than n take the full-width kernels (round-2 path) and still mix with pair rows
        wide = [v + n * (i % 3) for i, v in enumerate(m2)]
        bw = R.up(wide, 2 * nw)
        cw = R.op(L.pgpu_batch_encrypt, pk._h, bw, br2, rb)
        assert R.down(cw) == oc2
        assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c1, cw)) == osum
        assert R.down(R.op(L.pgpu_batch_ct_add_plain, pk._h, c1, bw)) == [a * ((1 + n * b) % nsq) % nsq for a, b in zip(oc1, wide)]
    finally:
        R.close()


def test_pair_rows_can_be_switched_off(engine):
    """PGPU_PAIR_ROWS=0 keeps the round-2 representation (Montgomery-form words): same results (own process: the switch is
    read once)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "pool_worker.py"), "batch_chain", "1"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PGPU_PAIR_ROWS="0", PGPU_MIN_SHARD="8"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert all(res["ok"].values()), res


def test_masked_table_gather_is_bit_identical(engine):
    """pgpu_set_table_gather_policy(1): the split-form kernels read every window-table entry and select (the address stream
    no longer depends on exponent digits).  CRT decrypt (both exponent policies), CT x PT with per-element exponents and
    the key-less seam modulo p^2 give the same bits as with indexed access."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048, True)
    n = p * q
    nsq = n * n
    rng = random.Random(991)
    count = 150
    m = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    e = [rng.getrandbits(64) for _ in range(count)]
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    oc = opk.encrypt(m, r)
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    assert L.pgpu_get_table_gather_policy() == 0
    _capi.check(L.pgpu_set_table_gather_policy(1))
    try:
        assert sk.decrypt(oc) == m
        bc, be = R.up(oc, 64), R.up(e, 1)
        assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, bc, be, 64)) == [pow(a, b, nsq) for a, b in zip(oc, e)]
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, bc)) == m
        psq = p * p
        base = [c % psq for c in oc[:40]]
        assert engine.mod_exp(base, [p - 1] * 40, psq) == [pow(c, p - 1, psq) for c in base]
        old = L.pgpu_get_secret_exponent_policy()
        _capi.check(L.pgpu_set_secret_exponent_policy(1))
        try:
            assert sk.decrypt(oc) == m
        finally:
            _capi.check(L.pgpu_set_secret_exponent_policy(old))
    finally:
        _capi.check(L.pgpu_set_table_gather_policy(0))
        R.close()


def test_two_batch_lanes(engine):
    """Two chains of resident batches on the two batch lanes of the GPU, issued interleaved; a result inherits the lane of
    its first operand; an operation whose operands live on different lanes is ordered by events and still correct."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048, True)
    n = p * q
    nsq = n * n
    rng = random.Random(4242)
    count = 300
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    R = Res()
    try:
        data, h = [], []
        for lane in (0, 1):
            m = [rng.randrange(n) for _ in range(count)]
            r = [rng.getrandbits(1024) for _ in range(count)]
            _capi.check(L.pgpu_set_batch_lane(lane))
            h.append((R.up(m, 32), R.up(r, 16)))
            data.append((m, r))
        _capi.check(L.pgpu_set_batch_lane(0))
        cs, ds = [None, None], [None, None]
        for rep in range(3):                      # interleaved issue: lane 0, lane 1, lane 0, ...
            for lane in (0, 1):
                cs[lane] = R.op(L.pgpu_batch_encrypt, pk._h, h[lane][0], h[lane][1], 1024)
                ds[lane] = R.op(L.pgpu_batch_decrypt_crt, sk._h, cs[lane])
        for lane in (0, 1):
            assert L.pgpu_batch_lane(cs[lane]) == lane and L.pgpu_batch_lane(ds[lane]) == lane
            assert R.down(ds[lane]) == data[lane][0]
            assert R.down(cs[lane]) == opk.encrypt(*data[lane])
        x = R.op(L.pgpu_batch_ct_add, pk._h, cs[0], cs[1])          # operands of both lanes
        assert L.pgpu_batch_lane(x) == 0
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) == [(a + b) % n for a, b in zip(data[0][0], data[1][0])]
        y = R.op(L.pgpu_batch_ct_add_plain, pk._h, cs[1], h[0][0])   # ciphertext of lane 1, plaintext of lane 0
        assert L.pgpu_batch_lane(y) == 1
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, y)) == [(a + b) % n for a, b in zip(data[1][0], data[0][0])]
    finally:
        _capi.check(L.pgpu_set_batch_lane(0))
        R.close()


@pytest.mark.parametrize("bits,count", [(3072, 300), (2048, 515), (1024, 1100)])
def test_sequential_halves_decrypt_kernel_is_bit_identical(engine, bits, count):
    """csrc/hensel_seq.hpp: both halves of a residue in the same lanes, one after the other (the form large launches take
    by default; forced here at test sizes).  Same plaintexts as the paired kernel and the oracle, ragged batch, with and
    without the masked table gather."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits, True)
    n = p * q
    nw = bits // 64
    rng = random.Random(bits + 5)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = [rng.getrandbits(bits // 2) for _ in range(count)]
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    try:
        c = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), bits // 2)
        assert L.pgpu_batch_row_limbs(c) > 0
        L.pgpu_debug_set_seq_decrypt(0)
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m       # the paired kernel
        L.pgpu_debug_set_seq_decrypt(2)
        try:
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m
            s2 = R.op(L.pgpu_batch_ct_add, pk._h, c, c)                     # a ciphertext that is not a fresh encryption
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, s2)) == [2 * v % n for v in m]
            _capi.check(L.pgpu_set_table_gather_policy(1))
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_seq_decrypt(4)                     # (back to the default: adaptive)
    finally:
        R.close()


@pytest.mark.parametrize("bits,ebits,count", [(2048, 40, 530), (2048, 2048, 70), (3072, 33, 130), (1024, 64, 1000)])
def test_sequential_halves_ct_times_pt_is_bit_identical(engine, bits, ebits, count):
    """hensel_modexp_seq_kernel (csrc/hensel_seq.hpp): CT x PT of a resident batch with both halves of a residue in the
    same lanes (the form launches of 16384+ elements take; forced here).  Same ciphertexts as pow() and as the paired
    kernel, ragged batch, edge exponents, a one-row exponent batch, with and without the masked table gather."""
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits, True)
    n = p * q
    nsq = n * n
    nw = bits // 64
    ew = (ebits + 63) // 64
    rng = random.Random(ebits + count)
    m = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(bits // 2) for _ in range(count)]
    e = ([0, 1, 2, (1 << ebits) - 1] + [rng.getrandbits(ebits) for _ in range(count)])[:count]
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    try:
        c = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), bits // 2)
        assert L.pgpu_batch_row_limbs(c) > 0
        oc = R.down(c)
        want = [pow(a, b, nsq) for a, b in zip(oc, e)]
        be = R.up(e, ew)
        L.pgpu_debug_set_seq_decrypt(0)
        assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c, be, ebits)) == want       # the paired kernel
        L.pgpu_debug_set_seq_decrypt(2)
        try:
            t = R.op(L.pgpu_batch_ct_mul, pk._h, c, be, ebits)
            assert L.pgpu_batch_row_limbs(t) > 0
            assert R.down(t) == want
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, t)) == [a * b % n for a, b in zip(m, e)]
            e1 = R.up([e[-1]], ew)
            assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c, e1, ebits)) == [pow(a, e[-1], nsq) for a in oc]
            _capi.check(L.pgpu_set_table_gather_policy(1))
            assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c, be, ebits)) == want
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_seq_decrypt(4)                     # (back to the default: adaptive)
    finally:
        R.close()


@pytest.mark.parametrize("bits,count", [(2048, 523), (3072, 140), (1024, 1030)])
def test_sequential_halves_ct_plus_ct_is_bit_identical(engine, bits, count):
    """pair_mul_seq_kernel (csrc/hensel_seq.hpp): CT + CT of resident batches as one pair product with both halves of a
    residue in the same lanes (launches of 16384+ elements; forced here): same ciphertexts as the product modulo n^2 and
    as the paired kernel, ragged batch, a one-row operand, chained twice."""
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits, True)
    n = p * q
    nsq = n * n
    nw = bits // 64
    rng = random.Random(77)
    m1 = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    m2 = [rng.randrange(n) for _ in range(count)]
    r1 = [rng.getrandbits(bits // 2) for _ in range(count)]
    r2 = [rng.getrandbits(bits // 2) for _ in range(count)]
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m1, nw), R.up(r1, nw // 2), bits // 2)
        c2 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m2, nw), R.up(r2, nw // 2), bits // 2)
        o1, o2 = R.down(c1), R.down(c2)
        want = [a * b % nsq for a, b in zip(o1, o2)]
        L.pgpu_debug_set_seq_decrypt(0)
        assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)) == want
        L.pgpu_debug_set_seq_decrypt(2)
        try:
            s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
            assert R.down(s) == want
            s2 = R.op(L.pgpu_batch_ct_add, pk._h, s, c1)
            assert R.down(s2) == [a * b % nsq for a, b in zip(want, o1)]
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, s2)) == [(2 * a + b) % n for a, b in zip(m1, m2)]
            one = R.up([o2[0]], 2 * nw)
            assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c1, one)) == [a * o2[0] % nsq for a in o1]
        finally:
            L.pgpu_debug_set_seq_decrypt(4)                     # (back to the default: adaptive)
    finally:
        R.close()


@pytest.mark.parametrize("bits,count", [(2048, 521), (3072, 150), (1024, 1050)])
def test_sequential_halves_djn_encrypt_is_bit_identical(engine, bits, count):
    """hensel_fb_encrypt_seq_kernel (csrc/hensel_seq.hpp): DJN encrypt onto pair rows with both halves of a residue in the
    same lanes (launches of 16384+ / 8192+ elements; forced here).  Same ciphertexts as the oracle and as the paired kernel:
    ragged batch, edge plaintexts and randomness."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits, True)
    n = p * q
    nw = bits // 64
    rng = random.Random(bits + count)
    m = ([0, 1, n - 1, n, (1 << (64 * nw)) - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = ([0, 1, (1 << (bits // 2)) - 1] + [rng.getrandbits(bits // 2) for _ in range(count)])[:count]
    opk = orc.PublicKey(n, bits)
    opk.set_djn(hs)
    want = opk.encrypt(m, r)
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    try:
        bm, br = R.up(m, nw), R.up(r, nw // 2)
        L.pgpu_debug_set_seq_decrypt(0)
        c0 = R.op(L.pgpu_batch_encrypt, pk._h, bm, br, bits // 2)
        assert L.pgpu_batch_row_limbs(c0) > 0 and R.down(c0) == want
        L.pgpu_debug_set_seq_decrypt(2)
        try:
            c = R.op(L.pgpu_batch_encrypt, pk._h, bm, br, bits // 2)
            assert L.pgpu_batch_row_limbs(c) > 0
            assert R.down(c) == want
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == [v % n for v in m]
        finally:
            L.pgpu_debug_set_seq_decrypt(4)                     # (back to the default: adaptive)
    finally:
        R.close()
