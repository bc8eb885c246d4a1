"""Round 5 additions, all through the C-ABI against the oracle (reference files cited per test):
  * the one-lane product-scanning CRT-decrypt kernel (csrc/hensel_ps.hpp: 2048-bit keys, 38 limbs of 28 bits per half):
    bit-identical with the default kernels and the oracle -- the two half-width exponentiations of
    PrivateKey::decryptCRT, ipcl/pri_key.cpp:114-146."""
import ctypes
import random

import pytest

from test_gpu_round4 import Res, key_case

pytestmark = pytest.mark.gpu


def _c_oracle_decrypt(p, q, bits, c_words):
    """CRT decrypt of EVERY row by the C oracle (fastest backend of the host, all cores): plaintext rows as 64-bit words"""
    from oracle import c_oracle
    from test_gpu_sha256_fullsize import key_limbs
    c_oracle.set_threads(min(c_oracle.lib().orc_max_threads(), c_oracle.usable_cpus()))
    be = c_oracle.ifma_modexp_batch if c_oracle.ifma_lib() is not None else (
        c_oracle.openssl_modexp_batch if c_oracle.openssl_lib() is not None else c_oracle.modexp_batch)
    _, _, sk_l = key_limbs(p, q, None, bits)
    return c_oracle.paillier_decrypt_crt_with(be, *sk_l, c_words)


def _raw_rows(n, bits, count, seed):
    """`count` values below n^2 that are NOT encryptions (L_p(c^(p-1)) has no structure), incl. n^2 - 1, 1 and n^2 - 2"""
    import numpy as np
    from pailliercryptolib_amd.limbs import ints_to_limbs
    nw = bits // 64
    rng = np.random.default_rng(seed)
    raw = np.frombuffer(rng.bytes(count * 2 * nw * 8), dtype=np.uint64).reshape(count, 2 * nw).copy()
    raw[:, -1] >>= np.uint64(4)                      # below 2^(2 bits - 4) < n^2 (n has its top bit set)
    edge = ints_to_limbs([n * n - 1, 1, n * n - 2], 2 * nw)
    raw[0], raw[count // 2], raw[count - 1] = edge[0], edge[1], edge[2]
    return raw


@pytest.mark.parametrize("bits,count", [(2048, 1), (2048, 63), (2048, 65), (2048, 300), (2048, 4100),
                                        (3072, 1), (3072, 65), (3072, 300), (3072, 1100),
                                        (1024, 1), (1024, 65), (1024, 300), (1024, 4100)])
def test_ps_decrypt_kernel_is_bit_identical(engine, bits, count):
    """csrc/hensel_ps.hpp: a whole half-width exponentiation per lane by product scanning (2048-bit keys: 38 limbs of 28 bits
    per half, 3072-bit keys: 56, 1024-bit keys: 19 limbs of 29 bits; by default from
    32768 ciphertexts up or beside busy neighbour lanes, forced here): same plaintexts as the default kernels and the
    oracle, ragged batches (padding lanes of the last wavefront), resident ciphertexts from every producer (encrypt,
    CT+CT, CT x PT, uploaded words -- relaxed limbs of the multi-lane kernels and canonical ones), edge plaintexts, and with
    the masked table gather."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nw = bits // 64
    rng = random.Random(count)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    m2 = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(bits // 2) for _ in range(count)]
    e = [rng.getrandbits(33) for _ in range(count)]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    R = Res()
    L = R.L
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), bits // 2)
        c2 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m2, nw), R.up(r, nw // 2), bits // 2)
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
        t = R.op(L.pgpu_batch_ct_mul, pk._h, s, R.up(e, 1), 33)
        raw = [rng.randrange(1, n * n) for _ in range(count)]        # not encryptions: L_p(c^(p-1)) has no structure
        raw[0] = n * n - 1
        up = R.up(raw, 2 * nw)
        up_pair = R.op(L.pgpu_batch_ct_add, pk._h, up, R.up([1], 2 * nw))   # the same values as pair rows
        want = [R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) for x in (c1, s, t, up_pair)]
        assert want[0] == m and want[1] == [(a + b) % n for a, b in zip(m, m2)]
        assert want[2] == [((a + b) * x) % n for a, b, x in zip(m, m2, e)]
        idx = sorted({0, count // 2, count - 1})
        assert [want[3][i] for i in idx] == osk.decrypt([raw[i] for i in idx])
        L.pgpu_debug_set_ps_decrypt(2)
        try:
            split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _capi.check(L.pgpu_decrypt_kernel_form(sk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
            assert (split.value, lanes.value, limbs.value) == (4, 1, {1024: 19, 2048: 38, 3072: 56}[bits])
            got = [R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) for x in (c1, s, t, up_pair, up)]
            assert got[:4] == want
            assert got[4] == want[3]          # word ciphertexts reach the kernel through the pair-row conversion
            _capi.check(L.pgpu_set_table_gather_policy(1))
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, t)) == want[2]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_ps_decrypt(1)
    finally:
        R.close()


def test_four_api_threads_take_the_quarter_chip_forms(engine):
    """Four host threads calling ipcl::PublicKey::encrypt / PrivateKey::decrypt on vectors of 8192 side by side (the
    reference's BM_Encrypt / BM_Decrypt shape from an OpenMP team, benchmark/bench_cryptography.cpp:24-63): their
    round-robin lanes see three active neighbours, so the launches take the quarter-chip forms (one-lane decrypt with a CU
    claim, encrypt and CRT kernel with the 80 000-byte claim) from four launching threads at once -- the configuration in
    which changing a kernel's attributes beside another thread's launch crashed inside the HIP runtime.  Every thread's
    round trip must hold, several times over."""
    import json
    import subprocess
    from pailliercryptolib_amd import build
    exe = build.build_api_bench()
    for _ in range(3):
        r = subprocess.run([exe, "--threads", "4", "8192", "3"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        assert json.loads(line)["round_trip_ok"] is True


def test_four_host_array_callers_take_the_quarter_chip_forms(engine):
    """Four threads calling pgpu_paillier_encrypt / pgpu_paillier_decrypt_crt on host arrays of their own (the C-ABI the
    reference's mod_exp.cpp seam binds; the reference's tests call encrypt / decrypt from four
    OpenMP threads, test_cryptography.cpp:45-57).  Each caller sees three others, so its launches take the quarter-chip
    forms -- sequential-halves encrypt onto pair rows and back to words, word ciphertexts converted to pair rows for the
    one-lane product-scanning decrypt -- and must deliver what a lone caller gets (full-chip paired kernels): ciphertexts
    bit-identical (the randomness is given), plaintexts back."""
    import threading
    import numpy as np
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    p, q, hs = key_case(2048)
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    B, nw, pw = 8192, 32, 16
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    sets = []
    for t in range(4):
        rng = np.random.default_rng(100 + t)
        m = np.frombuffer(rng.bytes(B * nw * 8), dtype=np.uint64).reshape(B, nw).copy()
        m[:, -1] &= np.uint64((1 << 62) - 1)
        r = np.frombuffer(rng.bytes(B * pw * 8), dtype=np.uint64).reshape(B, pw).copy()
        sets.append((m, r, np.empty((B, 2 * nw), dtype=np.uint64), np.empty((B, nw), dtype=np.uint64)))
    # the lone caller's ciphertexts first
    want = []
    for m, r, c, d in sets:
        _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(m), nw, nw, ptr(r), pw, pw, 64 * pw, ptr(c), B))
        want.append(c.copy())
    errs = []
    bar = threading.Barrier(4)

    def caller(k):
        m, r, c, d = sets[k]
        try:
            bar.wait()
            for _ in range(4):          # (the first round starts with idle neighbours: the later ones run in the quarter-chip mode)
                c[:] = 0
                d[:] = 0
                _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(m), nw, nw, ptr(r), pw, pw, 64 * pw, ptr(c), B))
                _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(c), ptr(d), B))
                if not np.array_equal(c, want[k]):
                    errs.append("caller %d: ciphertexts differ from the lone caller's" % k)
                if not np.array_equal(d, m):
                    errs.append("caller %d: round trip failed" % k)
        except Exception as e:                              # noqa: BLE001
            errs.append(repr(e))
            bar.abort()
    th = [threading.Thread(target=caller, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]


@pytest.mark.parametrize("bits,limbs", [(1024, 19), (3072, 56)])
def test_four_lanes_take_the_product_scanning_form_for_other_key_sizes(engine, bits, limbs):
    """Four resident batches of 8192 in flight under a 1024- / 3072-bit key (the headline's shape, BASELINE configs[1] + [2],
    at the other key sizes the reference's tests cover: test_cryptography.cpp runs 1024- and 2048-bit keys): beside three busy
    lanes every decrypt takes hensel_decrypt_ps_kernel<19,29,2> / <56,28,1> with its whole-CU claim (dynamic LDS on top of the
    kernel's own parking area).  Every lane's round trip must hold over several rounds, and the policy must report the form."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nw, count = bits // 64, 8192
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    L = _capi.lib()
    split, lanes, lb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_decrypt_kernel_form_ex(sk._h, count, 3, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(lb)))
    assert (split.value, lanes.value, lb.value) == (4, 1, limbs)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    live, sets = [], []

    def track(h):
        live.append(h)
        return h

    def up(a):
        h = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_upload(ptr(a), a.shape[0], a.shape[1], a.shape[1], ctypes.byref(h)))
        return track(h)

    def op(fn, *a):
        h = ctypes.c_void_p()
        _capi.check(fn(*a, ctypes.byref(h)))
        return track(h)
    try:
        for ln in range(4):
            rng = np.random.default_rng(bits + ln)
            m = np.frombuffer(rng.bytes(count * nw * 8), dtype=np.uint64).reshape(count, nw).copy()
            m[:, -1] &= np.uint64((1 << 62) - 1)
            r = np.frombuffer(rng.bytes(count * nw * 4), dtype=np.uint64).reshape(count, nw // 2).copy()
            _capi.check(L.pgpu_set_batch_lane(ln))
            sets.append((m, up(m), up(r)))
        outs, cts, raws, raw_outs = [None] * 4, [None] * 4, [], [None] * 4
        for ln in range(4):                    # a batch of NON-encryptions per lane (n^2 - 1, 1, n^2 - 2 among them)
            _capi.check(L.pgpu_set_batch_lane(ln))
            raw = _raw_rows(n, bits, count, 7000 + bits + ln)
            raws.append((raw, up(raw)))
        for _ in range(4):                     # (the first round starts beside idle lanes: the later ones run the quarter-chip forms)
            for ln in range(4):
                _capi.check(L.pgpu_set_batch_lane(ln))
                cts[ln] = op(L.pgpu_batch_encrypt, pk._h, sets[ln][1], sets[ln][2], bits // 2)
                outs[ln] = op(L.pgpu_batch_decrypt_crt, sk._h, cts[ln])
        for ln in range(4):                    # the same four lanes busy with the raw batches: quarter-chip launches again
            _capi.check(L.pgpu_set_batch_lane(ln))
            raw_outs[ln] = op(L.pgpu_batch_decrypt_crt, sk._h, raws[ln][1])
        _capi.check(L.pgpu_set_batch_lane(0))
        _capi.check(L.pgpu_synchronize())
        from oracle import paillier_oracle as orc
        from pailliercryptolib_amd.limbs import limbs_to_ints
        opk, osk = orc.PublicKey(n, bits), orc.PrivateKey(n, p, q)
        opk.set_djn(hs)
        rows = [0, count // 2, count - 1]
        for ln in range(4):
            got = np.empty((count, nw), dtype=np.uint64)
            _capi.check(L.pgpu_batch_download(outs[ln], ptr(got)))
            assert np.array_equal(got, sets[ln][0]), "lane %d: round trip failed" % ln
            # oracle level, not only the round trip: ciphertext rows against the Python oracle's encrypt ...
            cw = np.empty((count, 2 * nw), dtype=np.uint64)
            _capi.check(L.pgpu_batch_download(cts[ln], ptr(cw)))
            r_rows = np.empty((count, nw // 2), dtype=np.uint64)
            _capi.check(L.pgpu_batch_download(sets[ln][2], ptr(r_rows)))
            assert limbs_to_ints(cw[rows]) == opk.encrypt(limbs_to_ints(sets[ln][0][rows]), limbs_to_ints(r_rows[rows])), \
                "lane %d: ciphertext rows differ from the oracle" % ln
            # ... and the decrypt of non-encryptions: every row against the C oracle, the edge rows against pow()
            _capi.check(L.pgpu_batch_download(raw_outs[ln], ptr(got)))
            assert np.array_equal(got, _c_oracle_decrypt(p, q, bits, raws[ln][0])), "lane %d: raw decrypt differs from the C oracle" % ln
            assert limbs_to_ints(got[rows]) == osk.decrypt(limbs_to_ints(raws[ln][0][rows]))
    finally:
        _capi.check(L.pgpu_set_batch_lane(0))
        for h in live:
            L.pgpu_batch_destroy(h)


def test_ps_decrypt_at_the_tightest_headroom(engine):
    """The 56-limb product-scanning decrypt keeps R = 2^1568 >= 16 P only (csrc/capi_keys.inc: build_hensel; bounds:
    tests/test_hensel_model.py).  tests/golden/primes_worst_headroom.json holds primes p, q == 1 (mod 2^28) just below 2^1536:
    their multiplier k = -p^-1 mod 2^28 is 2^28 - 1, P = p k just below 2^1564 -- R / P = 16.0000001, the tightest any 3072-bit
    key gets.  Forced kernel against the oracle (pow) on raw values incl. n^2 - 1 and 1, and a round trip of 2100 encryptions
    (the reference's decryptCRT, ipcl/pri_key.cpp:114-146)."""
    import json
    import os
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "primes_worst_headroom.json")))
    p, q = int(k["p"], 16), int(k["q"], 16)
    assert (-pow(p, -1, 1 << 28)) % (1 << 28) == (1 << 28) - 1 and p.bit_length() == 1536
    n = p * q
    bits, nw, count = 3072, 48, 2100
    rng = random.Random(56)
    pk, sk = engine.PublicKey(n, bits), engine.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    R = Res()
    L = R.L
    L.pgpu_debug_set_ps_decrypt(2)
    try:
        split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.check(L.pgpu_decrypt_kernel_form(sk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
        assert (split.value, lanes.value, limbs.value) == (4, 1, 56)
        raw = [n * n - 1, 1, n * n - 2, (1 << 6143) + 1] + [rng.randrange(1, n * n) for _ in range(40)]
        got = R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, R.up(raw, 2 * nw)))
        assert got == osk.decrypt(raw)
        m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
        r = [rng.randrange(1, n) for _ in range(count)]
        c = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw), bits)
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m
        s = R.op(L.pgpu_batch_ct_add, pk._h, c, c)
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, s)) == [(2 * x) % n for x in m]
    finally:
        L.pgpu_debug_set_ps_decrypt(1)
        R.close()


@pytest.mark.parametrize("bits,count", [(2048, 20000), (2048, 32768 + 700), (1024, 32768 + 9000), (3072, 32768 + 300)])
def test_lone_decrypts_in_rounds_of_the_product_scanning_form(engine, bits, count):
    """A lone CRT decrypt of more than 16384 / 24576 ciphertexts (2048- / 3072-bit keys) runs in rounds of 32768 of the
    one-lane product-scanning kernel; a mostly empty last round is cut off as a launch of its own in the form its size takes
    (csrc/policy.hpp: ps_min_count, ps_split_head; capi.cpp: decrypt_on).  Resident pair rows and word ciphertexts from host
    arrays (the pair-row conversion of each part) must both give the plaintexts back -- PrivateKey::decrypt on a vector of any
    size, ipcl/pri_key.cpp:65-112."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nw = bits // 64
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    L = _capi.lib()
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_decrypt_kernel_form(sk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    assert split.value == 4 and lanes.value == 1
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(count)
    m = np.frombuffer(rng.bytes(count * nw * 8), dtype=np.uint64).reshape(count, nw).copy()
    m[:, -1] &= np.uint64((1 << 62) - 1)
    m[0] = 0
    r = np.frombuffer(rng.bytes(count * nw * 4), dtype=np.uint64).reshape(count, nw // 2).copy()
    live = []

    def op(fn, *a):
        h = ctypes.c_void_p()
        _capi.check(fn(*a, ctypes.byref(h)))
        live.append(h)
        return h
    try:
        hm = op(L.pgpu_batch_upload, ptr(m), count, nw, nw)
        hr = op(L.pgpu_batch_upload, ptr(r), count, nw // 2, nw // 2)
        c = op(L.pgpu_batch_encrypt, pk._h, hm, hr, bits // 2)
        d = op(L.pgpu_batch_decrypt_crt, sk._h, c)
        got = np.empty((count, nw), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(d, ptr(got)))
        assert np.array_equal(got, m)
        cw = np.empty((count, 2 * nw), dtype=np.uint64)          # the same ciphertexts as canonical words, from a host array
        _capi.check(L.pgpu_batch_download(c, ptr(cw)))
        got[:] = 0
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(cw), ptr(got), count))
        assert np.array_equal(got, m)
        # Oracle level, not only round trips (the partial-round split -- policy.hpp: ps_split_head -- is launch logic of its
        # own): three ciphertext rows of every launch part (rounds of 32768, then the tail) against the Python oracle ...
        from oracle import paillier_oracle as orc
        from pailliercryptolib_amd.limbs import limbs_to_ints
        opk, osk = orc.PublicKey(n, bits), orc.PrivateKey(n, p, q)
        opk.set_djn(hs)
        rows = sorted({i for lo in range(0, count, 32768) for i in (lo, (lo + min(lo + 32768, count)) // 2, min(lo + 32768, count) - 1)})
        assert limbs_to_ints(cw[rows]) == opk.encrypt(limbs_to_ints(m[rows]), limbs_to_ints(r[rows])), "ciphertext rows differ from the oracle"
        # ... and a batch of NON-encryptions of the same size (n^2 - 1, 1, n^2 - 2 among them), resident and from host arrays:
        # every row against the C oracle, the rows at the part boundaries against pow()
        raw = _raw_rows(n, bits, count, count + bits)
        want = _c_oracle_decrypt(p, q, bits, raw)
        edge = sorted(set(rows) | {count // 2})
        assert limbs_to_ints(want[edge]) == osk.decrypt(limbs_to_ints(raw[edge]))         # (the checker itself, on the edge rows)
        hraw = op(L.pgpu_batch_upload, ptr(raw), count, 2 * nw, 2 * nw)
        draw = op(L.pgpu_batch_decrypt_crt, sk._h, hraw)
        _capi.check(L.pgpu_batch_download(draw, ptr(got)))
        assert np.array_equal(got, want), "resident raw decrypt differs from the C oracle"
        got[:] = 0
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(raw), ptr(got), count))
        assert np.array_equal(got, want), "host-array raw decrypt differs from the C oracle"
    finally:
        for h in live:
            L.pgpu_batch_destroy(h)
