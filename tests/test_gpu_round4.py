"""Round 4 additions, all through the C-ABI against the oracle (reference files cited per test):
  * the masked fixed-base product of DJN encrypt (pgpu_set_table_gather_policy(1): pub_key.cpp:51-64 with the
    constant-address table access of mbx_exp_mb8, mod_exp.cpp:508-516) -- every compiled kernel form, resident and from
    host arrays;
  * four batch lanes with the adaptive kernel-form policy: interleaved chains on all lanes at ragged sizes on both sides
    of every form threshold (2049 / 2100 / 4097 / 8192 + 1: the sizes of benchmark/bench_cryptography.cpp:10-19 and
    their neighbours), whatever mix of paired / sequential-halves / CU-claiming launches the probe picks;
  * pinned host blocks (pgpu_host_alloc) as sources / targets of every transfer."""
import ctypes
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def key_case(bits):
    if bits == 2048:
        k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
        return int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
    c = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == bits and c["djn"]][0]
    return int(c["p"], 16), int(c["q"], 16), int(c["hs"], 16)


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


@pytest.mark.parametrize("bits,count", [(2048, 37), (2048, 700), (2048, 9001), (2048, 17000), (1024, 300), (1024, 40000),
                                        (3072, 70), (3072, 9000)])
def test_masked_fixed_base_encrypt(engine, bits, count):
    """hs^r through the small-window table with every entry read and selected: bit-identical with the indexed product and
    with the oracle (sampled: an 8192-element oracle pass takes minutes), pair rows and host words, edge randomness."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nw, pw = bits // 64, bits // 128
    rng = random.Random(bits + count)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = ([0, 1, (1 << (bits // 2)) - 1] + [rng.getrandbits(bits // 2) for _ in range(count)])[:count]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, bits)
    opk.set_djn(hs)
    R = Res()
    L = R.L
    try:
        bm, br = R.up(m, nw), R.up(r, pw)
        indexed = R.down(R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 64 * pw))
        _capi.check(L.pgpu_set_table_gather_policy(1))
        try:
            c = R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 64 * pw)
            masked = R.down(c)
            assert masked == indexed
            idx = sorted({0, 1, 2, count // 2, count - 1, rng.randrange(count), rng.randrange(count)} & set(range(count)))
            assert [masked[i] for i in idx] == opk.encrypt([m[i] for i in idx], [r[i] for i in idx])
            # the masked decrypt reads it back (both halves of the path under the policy)
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m
            # host arrays in, plain words out (the full exit of the kernels)
            k = min(count, 1200)
            ma, ra = R.i2l(m[:k], nw), R.i2l(r[:k], pw)
            out = np.empty((k, 2 * nw), dtype=np.uint64)
            _capi.check(L.pgpu_paillier_encrypt(pk._h, ma.ctypes.data_as(ctypes.c_void_p), nw, nw,
                                                ra.ctypes.data_as(ctypes.c_void_p), pw, pw, 64 * pw,
                                                out.ctypes.data_as(ctypes.c_void_p), k))
            assert R.l2i(out) == indexed[:k]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
        # back on the indexed table: nothing was rebuilt or mixed up
        assert R.down(R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 64 * pw)) == indexed
    finally:
        R.close()


def test_masked_fixed_base_full_width_kernel(engine):
    """plaintext rows WIDER than n take the full-width fixed-base kernel (kernels.hpp: fb_encrypt_kernel): its masked table
    access against the oracle as well"""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048)
    n = p * q
    rng = random.Random(5)
    count = 130
    m = [rng.randrange(n) + n * (i % 3) for i in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    pk = engine.PublicKey(n, 2048, hs=hs)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    R = Res()
    L = R.L
    try:
        bm, br = R.up(m, 64), R.up(r, 16)
        want = opk.encrypt([v % n for v in m], r)
        _capi.check(L.pgpu_set_table_gather_policy(1))
        try:
            assert R.down(R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 1024)) == want
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
        assert R.down(R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 1024)) == want
    finally:
        R.close()


@pytest.mark.parametrize("policy", [4, 3, 1])
def test_four_lanes_ragged_sizes(engine, policy):
    """Chains on all four batch lanes issued interleaved, ragged sizes around every kernel-form threshold: the launcher's
    choice (paired / sequential halves / CU claim -- it depends on what the neighbour lanes are doing at that moment) never
    changes a result.  Every size decrypts to its plaintexts; sampled ciphertext rows against the oracle."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048)
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(hs)
    R = Res()
    L = R.L
    assert L.pgpu_batch_lanes() >= 4
    sizes = [2049, 2100, 4097, 8193, 16385, 513, 8192, 2048]
    rng = random.Random(policy)
    old = L.pgpu_debug_get_seq_decrypt()
    L.pgpu_debug_set_seq_decrypt(policy)
    try:
        jobs = []
        for i, cnt in enumerate(sizes):
            m = [rng.randrange(n) for _ in range(cnt)]
            r = [rng.getrandbits(1024) for _ in range(cnt)]
            _capi.check(L.pgpu_set_batch_lane(i % 4))
            jobs.append((m, r, R.up(m, 32), R.up(r, 16)))
        _capi.check(L.pgpu_set_batch_lane(0))
        cts, outs = [], []
        for rep in range(2):                     # two rounds: the second finds every lane busy
            for m, r, bm, br in jobs:
                cts.append(R.op(L.pgpu_batch_encrypt, pk._h, bm, br, 1024))
                outs.append(R.op(L.pgpu_batch_decrypt_crt, sk._h, cts[-1]))
        # sums across lanes: operands of different lanes are ordered in by events
        s = R.op(L.pgpu_batch_ct_add, pk._h, cts[6], cts[len(sizes) + 6])
        for k, o in enumerate(outs):
            m = jobs[k % len(sizes)][0]
            assert R.down(o) == m, (policy, k, len(m))
        for k in (0, 4, len(sizes) + 1):
            m, r = jobs[k % len(sizes)][0], jobs[k % len(sizes)][1]
            got = R.down(cts[k])
            idx = [0, len(m) // 2, len(m) - 1]
            assert [got[i] for i in idx] == opk.encrypt([m[i] for i in idx], [r[i] for i in idx])
        m6 = jobs[6][0]
        assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, s)) == [(2 * v) % n for v in m6]
    finally:
        L.pgpu_debug_set_seq_decrypt(old)
        _capi.check(L.pgpu_set_batch_lane(0))
        R.close()


def test_pinned_host_blocks(engine):
    """Buffers from pgpu_host_alloc are the DMA source / target of uploads, downloads and the host-pointer entry points;
    same results as pageable arrays; an upload that was only queued is complete after pgpu_host_wait."""
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    p, q, hs = key_case(2048)
    n = p * q
    L = _capi.lib()
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = random.Random(11)
    count = 3000
    m = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    m_np, r_np = ints_to_limbs(m, 32), ints_to_limbs(r, 16)

    def pinned(shape):
        nbytes = int(np.prod(shape)) * 8
        ptr = ctypes.c_void_p()
        _capi.check(L.pgpu_host_alloc(nbytes, ctypes.byref(ptr)))
        arr = np.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr.value), dtype=np.uint64).reshape(shape)
        return ptr, arr
    blocks = []
    try:
        pm, am = pinned((count, 32)); blocks.append(pm)
        pr, ar = pinned((count, 16)); blocks.append(pr)
        pc, ac = pinned((count, 64)); blocks.append(pc)
        pd, ad = pinned((count, 32)); blocks.append(pd)
        am[:], ar[:] = m_np, r_np
        # host-pointer entry points on pinned buffers vs pageable ones
        c_page = np.empty((count, 64), dtype=np.uint64)
        args = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _capi.check(L.pgpu_paillier_encrypt(pk._h, args(m_np), 32, 32, args(r_np), 16, 16, 1024, args(c_page), count))
        _capi.check(L.pgpu_paillier_encrypt(pk._h, pm, 32, 32, pr, 16, 16, 1024, pc, count))
        assert np.array_equal(ac, c_page)
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, pc, pd, count))
        assert np.array_equal(ad, m_np)
        # resident batches: upload from a block returns at once, the block is reusable after pgpu_host_wait
        hm, hr, hc, ho = (ctypes.c_void_p() for _ in range(4))
        _capi.check(L.pgpu_batch_upload(pm, count, 32, 32, ctypes.byref(hm)))
        _capi.check(L.pgpu_batch_upload(pr, count, 16, 16, ctypes.byref(hr)))
        _capi.check(L.pgpu_host_wait(pm))
        _capi.check(L.pgpu_host_wait(ctypes.c_void_p(pr.value + 4096)))       # any address inside the block
        am[:] = 0                                                              # (the upload has been consumed)
        _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, 1024, ctypes.byref(hc)))
        _capi.check(L.pgpu_batch_decrypt_crt(sk._h, hc, ctypes.byref(ho)))
        _capi.check(L.pgpu_batch_download(ho, pd))
        assert limbs_to_ints(ad) == m
        _capi.check(L.pgpu_batch_download(hc, pc))                             # pair rows -> words, straight into the block
        assert np.array_equal(ac, c_page)
        # a slice of a block (offset, shorter) is recognised as pinned as well; a pageable array still works
        half = count // 2
        sub = ctypes.c_void_p(pc.value + half * 64 * 8)
        hsub = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_upload(sub, count - half, 64, 64, ctypes.byref(hsub)))
        hout = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_decrypt_crt(sk._h, hsub, ctypes.byref(hout)))
        got = np.empty((count - half, 32), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(hout, args(got)))
        assert limbs_to_ints(got) == m[half:]
        for h in (hm, hr, hc, ho, hsub, hout):
            L.pgpu_batch_destroy(h)
    finally:
        for b in blocks:
            L.pgpu_host_free(b)
    assert L.pgpu_host_wait(ctypes.c_void_p(12345)) == 0     # not a block: nothing pending, no error


def test_small_transfers_through_the_bounce_buffer(engine):
    """uploads / downloads of at most 256 KB on a one-GPU pool run on the calling thread through its pinned bounce buffer:
    back-to-back uploads (the second has to wait for the first copy), pair-row and plain downloads"""
    p, q, hs = key_case(2048)
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    L = R.L
    rng = random.Random(3)
    try:
        for count in (1, 16, 100, 1000):        # 1000 x 32 words = 256 000 B: just inside
            m1 = [rng.randrange(n) for _ in range(count)]
            m2 = [rng.randrange(n) for _ in range(count)]
            r = [rng.getrandbits(1024) for _ in range(count)]
            b1, b2, br = R.up(m1, 32), R.up(m2, 32), R.up(r, 16)
            assert R.down(b1) == m1 and R.down(b2) == m2
            c1 = R.op(L.pgpu_batch_encrypt, pk._h, b1, br, 1024)
            c2 = R.op(L.pgpu_batch_encrypt, pk._h, b2, br, 1024)
            s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, s)) == [(a + b) % n for a, b in zip(m1, m2)]
            assert len(R.down(s)) == count
    finally:
        R.close()


@pytest.mark.parametrize("bits,count", [(2048, 8192), (2048, 2100), (2048, 20000), (1024, 40000), (3072, 4100)])
def test_host_array_callers_side_by_side(engine, bits, count):
    """Two threads calling pgpu_paillier_encrypt / pgpu_paillier_decrypt_crt on host arrays of their own (the reference's
    BM_Encrypt / BM_Decrypt shape, benchmark/bench_cryptography.cpp:24-63, from an OpenMP team): each sees the other
    through the lane activity stamps (PGPU_HOST_ADAPT=1, switched on here), so the launches take the part-chip forms of the
    adaptive policy -- word ciphertexts converted to pair rows on the way in, pair rows back to words on the way out; the
    larger counts take the conversion by size alone (sequential-halves / one-lane decrypt of word ciphertexts).  Bit-identical with the same calls
    issued alone under the fixed paired policy, and with the oracle (sampled)."""
    import threading
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    p, q, hs = key_case(bits)
    n = p * q
    nw, pw = bits // 64, bits // 128
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, bits)
    opk.set_djn(hs)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    sets = []
    for t in range(2):
        rng = random.Random(bits * 7 + count + t)
        m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
        r = ([0, 1, (1 << (bits // 2)) - 1] + [rng.getrandbits(bits // 2) for _ in range(count)])[:count]
        sets.append((m, r, ints_to_limbs(m, nw), ints_to_limbs(r, pw)))
    # reference run: one caller, lone-caller forms only
    L.pgpu_debug_set_seq_decrypt(1)
    alone = []
    try:
        for m, r, ma, ra in sets:
            c = np.empty((count, 2 * nw), dtype=np.uint64)
            d = np.empty((count, nw), dtype=np.uint64)
            _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(ma), nw, nw, ptr(ra), pw, pw, 64 * pw, ptr(c), count))
            _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(c), ptr(d), count))
            alone.append((c, d))
    finally:
        L.pgpu_debug_set_seq_decrypt(4)                 # the default: adaptive
    for (m, r, ma, ra), (c, d) in zip(sets, alone):
        assert limbs_to_ints(d) == m
        assert limbs_to_ints(c[:6]) == opk.encrypt(m[:6], r[:6])
    _capi.check(L.pgpu_set_timing(1))
    was = L.pgpu_debug_set_host_adapt(1)        # (PGPU_HOST_ADAPT: off by default -- it does not pay for synchronous callers)
    outs = [[], []]
    errs = []
    bar = threading.Barrier(2)

    def caller(t):
        m, r, ma, ra = sets[t]
        try:
            bar.wait()
            for _ in range(3):
                c = np.empty((count, 2 * nw), dtype=np.uint64)
                d = np.empty((count, nw), dtype=np.uint64)
                _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(ma), nw, nw, ptr(ra), pw, pw, 64 * pw, ptr(c), count))
                _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(c), ptr(d), count))
                outs[t].append((c, d))
        except Exception as e:                              # noqa: BLE001
            errs.append(repr(e))
            try:
                bar.abort()
            except Exception:                               # noqa: BLE001
                pass
    th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    mx = 256
    kinds, forms, ms = (ctypes.c_int * mx)(), (ctypes.c_int * mx)(), (ctypes.c_double * mx)()
    got = L.pgpu_timing_collect_ex(kinds, forms, ms, mx)
    L.pgpu_set_timing(0)
    L.pgpu_debug_set_host_adapt(was)
    assert not errs, errs
    for t in range(2):
        for c, d in outs[t]:
            assert np.array_equal(c, alone[t][0])
            assert np.array_equal(d, alone[t][1])
    # (which forms ran depends on how the two threads met; with 2048-bit keys and a batch that fills half the chip the
    # sequential-halves form must have been among them)
    if bits == 2048 and count == 8192:
        assert any(forms[i] & 2 for i in range(got)), [forms[i] for i in range(got)]


@pytest.mark.parametrize("pinned", [True, False])
def test_async_download_tickets(engine, pinned):
    """pgpu_batch_download_async / pgpu_ticket_wait: the download runs on a worker of the pool after the batch's kernels, the
    caller goes on queueing work on other lanes meanwhile.  Batches in every representation (plain words, Montgomery words
    or pair rows from encrypt, decrypt results), ragged sizes, pinned and pageable targets; equal to the synchronous
    download, and decrypt results equal to the plaintexts."""
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs
    L = _capi.lib()
    p, q, hs = key_case(2048)
    n = p * q
    nw, pw = 32, 16
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = random.Random(77)
    held, live, tickets = [], [], []

    def target(rows, words):
        if not pinned:
            return np.zeros((rows, words), dtype=np.uint64)
        pp = ctypes.c_void_p()
        _capi.check(L.pgpu_host_alloc(rows * words * 8, ctypes.byref(pp)))
        held.append(pp)
        a = np.frombuffer((ctypes.c_uint8 * (rows * words * 8)).from_address(pp.value), dtype=np.uint64).reshape(rows, words)
        a[:] = 0
        return a
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def op(fn, *a):
        h = ctypes.c_void_p()
        _capi.check(fn(*a, ctypes.byref(h)))
        live.append(h)
        return h
    try:
        jobs = []
        for lane, count in enumerate((37, 2100, 8192, 513)):
            _capi.check(L.pgpu_set_batch_lane(lane))
            m = ints_to_limbs([rng.randrange(n) for _ in range(count)], nw)
            r = ints_to_limbs([rng.getrandbits(1024) for _ in range(count)], pw)
            bm = op(L.pgpu_batch_upload, ptr(m), count, nw, nw)
            br = op(L.pgpu_batch_upload, ptr(r), count, pw, pw)
            c = op(L.pgpu_batch_encrypt, pk._h, bm, br, 1024)
            d = op(L.pgpu_batch_decrypt_crt, sk._h, c)
            for b, words, want in ((bm, nw, m), (c, 2 * nw, None), (d, nw, m)):
                out = target(count, words)
                t = ctypes.c_void_p()
                _capi.check(L.pgpu_batch_download_async(b, ptr(out), ctypes.byref(t)))
                jobs.append((t, b, out, words, want))
        _capi.check(L.pgpu_set_batch_lane(0))
        for t, b, out, words, want in jobs:
            _capi.check(L.pgpu_ticket_wait(t))
            ref = np.empty_like(out)
            _capi.check(L.pgpu_batch_download(b, ptr(ref)))
            assert np.array_equal(out, ref)
            if want is not None:
                assert np.array_equal(out, want)
        assert L.pgpu_batch_download_async(None, None, None) != 0      # null arguments are refused, no ticket is made
    finally:
        L.pgpu_set_batch_lane(0)
        for h in live:
            L.pgpu_batch_destroy(h)
        for pp in held:
            L.pgpu_host_free(pp)


def test_shared_operand_across_lanes_and_early_destroy(engine):
    """An operand shared by operations on other lanes (the ipcl:: layer's cached randomness, a ciphertext several threads
    add to): consumers wait on the operand's one 'produced' event, its own lane waits for the readers only when the operand
    is destroyed.  Here the operand is destroyed right after the readers were queued and its lane immediately reuses the
    memory for new uploads: every reader still saw the operand (CT+CT against the oracle's product), many rounds."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    p, q, hs = key_case(2048)
    n = p * q
    nsq = n * n
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = random.Random(4242)
    count = 3000
    R = Res()
    try:
        for rnd in range(6):
            xs = [rng.randrange(nsq) for _ in range(count)]
            _capi.check(L.pgpu_set_batch_lane(0))
            shared = R.up(xs, 64)                                   # lane 0
            outs = []
            for lane in (1, 2, 3):
                _capi.check(L.pgpu_set_batch_lane(lane))
                ys = [rng.randrange(nsq) for _ in range(count)]
                by = R.up(ys, 64)                                   # lane `lane`
                outs.append((ys, R.op(L.pgpu_batch_ct_add, pk._h, by, shared)))   # result on `lane`, reads lane 0's batch
            L.pgpu_batch_destroy(shared)                            # readers are merely queued
            R.live.remove(shared)
            _capi.check(L.pgpu_set_batch_lane(0))
            junk = [R.up([rng.randrange(nsq) for _ in range(count)], 64) for _ in range(3)]   # lane 0 recycles the memory
            for ys, o in outs:
                got = R.down(o)
                idx = [0, 1, count // 2, count - 1] + [rng.randrange(count) for _ in range(12)]
                assert [got[i] for i in idx] == [xs[i] * ys[i] % nsq for i in idx]
            assert len(junk) == 3
    finally:
        L.pgpu_set_batch_lane(0)
        R.close()
