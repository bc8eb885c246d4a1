"""Round 6, through the C-ABI against the oracle:
  * the LATENCY form of the CRT-decrypt exponentiation (csrc/hensel_wave.hpp: one exponentiation per wavefront, a limb per
    lane, between the one-lane entry and exit of the product-scanning kernel) -- what a lone decrypt of up to 512 ciphertexts
    takes by default: the reference's BM_Decrypt sizes 16 ... 512 (benchmark/bench_cryptography.cpp:10-19), the two half-width
    exponentiations of PrivateKey::decryptCRT (ipcl/pri_key.cpp:114-146)."""
import ctypes
import random

import pytest

from test_gpu_round4 import Res, key_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,count", [(2048, 1), (2048, 7), (2048, 16), (2048, 65), (2048, 300), (2048, 512), (2048, 1100),
                                        (3072, 1), (3072, 9), (3072, 130), (1024, 1), (1024, 16), (1024, 300)])
def test_wave_decrypt_kernel_is_bit_identical(engine, bits, count):
    """Forced on (also beyond its default range: 1100 ciphertexts are two wavefronts per SIMD) and off: same plaintexts as
    the multi-lane latency forms and the oracle -- ciphertexts from every producer (encrypt, CT+CT, CT x PT, uploaded words:
    relaxed and canonical rows), NON-encryptions incl. n^2 - 1 and 1 against pow(), edge plaintexts, masked table access."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nw = bits // 64
    rng = random.Random(7 * count + bits)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    m2 = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(bits // 2) for _ in range(count)]
    e = [rng.getrandbits(33) for _ in range(count)]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    R = Res()
    L = R.L
    form = lambda: tuple(v.value for v in _form(L, _capi, sk, count))
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), bits // 2)
        c2 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m2, nw), R.up(r, nw // 2), bits // 2)
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
        t = R.op(L.pgpu_batch_ct_mul, pk._h, s, R.up(e, 1), 33)
        raw = ([n * n - 1, 1, n * n - 2] + [rng.randrange(1, n * n) for _ in range(count)])[:count]
        up = R.up(raw, 2 * nw)
        srcs = (c1, s, t, up)
        L.pgpu_debug_set_wave_decrypt(0)
        try:
            assert form()[0] != 5
            want = [R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) for x in srcs]
            assert want[0] == m and want[1] == [(a + b) % n for a, b in zip(m, m2)]
            assert want[2] == [((a + b) * x) % n for a, b, x in zip(m, m2, e)]
            idx = sorted({0, 1, 2, count // 2, count - 1} & set(range(count)))
            assert [want[3][i] for i in idx] == osk.decrypt([raw[i] for i in idx])
            L.pgpu_debug_set_wave_decrypt(2)
            assert form() == (5, 64, {1024: 19, 2048: 38, 3072: 56}[bits])
            got = [R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) for x in srcs]
            assert got == want, "the wavefront-wide form differs from the multi-lane forms"
            _capi.check(L.pgpu_set_table_gather_policy(1))
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, t)) == want[2]
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, up)) == want[3]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_wave_decrypt(1)
        assert (form()[0] == 5) == (count <= 512)                 # the default: small lone launches
    finally:
        R.close()


def _form(L, _capi, sk, count):
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_decrypt_kernel_form(sk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    return split, lanes, limbs


def test_wave_decrypt_from_host_arrays_and_the_api(engine):
    """The host-pointer entry point (word ciphertexts through the pair-row conversion) and the python engine's decrypt at the
    reference's small benchmark sizes: whatever form the policy picks, plaintexts back; every row of a batch of
    non-encryptions against the C oracle."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from test_gpu_round5 import _c_oracle_decrypt, _raw_rows
    p, q, hs = key_case(2048)
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    L = _capi.lib()
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = random.Random(606)
    for count in (16, 64, 128, 256, 512):
        m = [rng.randrange(n) for _ in range(count)]
        r = [rng.getrandbits(1024) for _ in range(count)]
        assert sk.decrypt(pk.encrypt(m, r)) == m
        raw = _raw_rows(n, 2048, count, 9000 + count)
        got = np.zeros((count, 32), dtype=np.uint64)
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(raw), ptr(got), count))
        assert np.array_equal(got, _c_oracle_decrypt(p, q, 2048, raw)), "host-array decrypt of %d raw rows differs from the C oracle" % count


@pytest.mark.parametrize("bits,count,e_bits", [(2048, 1, 33), (2048, 7, 1024), (2048, 65, 64), (2048, 300, 17), (2048, 1024, 33),
                                               (2048, 1100, 5), (3072, 9, 40), (3072, 130, 1536), (1024, 16, 33), (1024, 300, 512)])
def test_wave_modexp_n2_kernel_is_bit_identical(engine, bits, count, e_bits):
    """csrc/hensel_wave_n2.hpp: CipherText * PlainText (ipcl/ciphertext.cpp:143-162) with one exponentiation per wavefront on
    pair rows -- what launches of up to 1024 resident ciphertexts take by default (the reference's BM_Mul_CTPT sizes,
    bench_ops.cpp:138-149, with its 1024-bit plaintexts; its tests use 32-bit ones, test_ops.cpp:294-325).  Forced on and
    off: pow(c, e, n^2) either way, for every producer of rows, edge exponents (0, 1, all ones) and bases (0, 1, n^2 - 1, n),
    one exponent for the whole batch, results re-read by CT+CT / CRT decrypt / the kernel itself, masked table access."""
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nsq = n * n
    nw, ew = bits // 64, (e_bits + 63) // 64
    rng = random.Random(count * 1000 + e_bits + bits)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = [rng.getrandbits(bits // 2) for _ in range(count)]
    e = ([0, 1, (1 << e_bits) - 1, 2] + [rng.getrandbits(e_bits) for _ in range(count)])[:count]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    L = R.L

    def form():
        split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.check(L.pgpu_modexp_n2_kernel_form(pk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
        return split.value, lanes.value, limbs.value
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), bits // 2)      # rows of the encrypt kernel
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c1)                                        # ... of CT + CT
        raw = ([0, 1, nsq - 1, n, n + 1] + [rng.randrange(nsq) for _ in range(count)])[:count]
        up = R.up(raw, 2 * nw)                                                               # ... uploaded words
        eh = R.up(e, ew)
        srcs = [c1, s, up]
        vals = [R.down(x) for x in srcs]
        want = [[pow(c, x, nsq) for c, x in zip(v, e)] for v in vals]
        L.pgpu_debug_set_wave_decrypt(0)
        try:
            assert form()[0] != 5
            assert [R.down(R.op(L.pgpu_batch_ct_mul, pk._h, x, eh, e_bits)) for x in srcs] == want, "the multi-lane kernels differ from pow()"
            L.pgpu_debug_set_wave_decrypt(2)
            assert form() == (5, 64, {1024: 38, 2048: 72, 3072: 112}[bits])
            outs = [R.op(L.pgpu_batch_ct_mul, pk._h, x, eh, e_bits) for x in srcs]
            assert [R.down(o) for o in outs] == want, "the wavefront-wide kernel differs from pow()"
            t2 = R.op(L.pgpu_batch_ct_mul, pk._h, outs[0], eh, e_bits)                       # its rows, read by itself,
            assert R.down(t2) == [pow(c, x, nsq) for c, x in zip(want[0], e)]
            one_e = R.up([e[-1]], ew)                                                        # one exponent for the whole batch
            assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c1, one_e, e_bits)) == [pow(c, e[-1], nsq) for c in vals[0]]
            if count <= 300:
                _capi.check(L.pgpu_set_table_gather_policy(1))
                assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, s, eh, e_bits)) == want[1]
                _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_wave_decrypt(0)                                                 # ... by the multi-lane kernels
            assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, outs[0], c1)) == [a * b % nsq for a, b in zip(want[0], vals[0])]
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, outs[0])) == [a * x % n for a, x in zip(m, e)]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_wave_decrypt(1)
        assert (form()[0] == 5) == (count <= 1024)
    finally:
        R.close()


@pytest.mark.parametrize("bits,count", [(2048, 1), (2048, 7), (2048, 65), (2048, 300), (2048, 1024), (2048, 1100),
                                        (3072, 9), (3072, 130), (1024, 16), (1024, 300)])
def test_wave_fb_encrypt_kernel_is_bit_identical(engine, bits, count):
    """csrc/hensel_wave_n2.hpp: hensel_fb_encrypt_wave_kernel -- DJN encrypt (ipcl/pub_key.cpp:51-64, 88-105) of up to 1024
    elements onto pair rows with one wavefront per element: the oracle's ciphertexts, forced on and off, edge plaintexts
    (0, 1, n - 1) and randomness (0, 1, all ones), masked table access; the rows decrypt and add like any others."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(bits)
    n = p * q
    nsq = n * n
    nw, rb = bits // 64, bits // 2
    rng = random.Random(31 * count + bits)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = ([0, 1, (1 << rb) - 1] + [rng.getrandbits(rb) for _ in range(count)])[:count]
    pk, sk = engine.PublicKey(n, bits, hs=hs), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, bits)
    opk.set_djn(hs)
    want = opk.encrypt(m, r)
    R = Res()
    L = R.L

    def form():
        split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.check(L.pgpu_encrypt_kernel_form_ex(pk._h, nw, count, -1, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))   # (-1: resident results)
        return split.value, lanes.value, limbs.value
    try:
        hm, hr = R.up(m, nw), R.up(r, nw // 2)
        L.pgpu_debug_set_wave_decrypt(0)
        try:
            assert form()[0] != 5
            assert R.down(R.op(L.pgpu_batch_encrypt, pk._h, hm, hr, rb)) == want, "the multi-lane kernels differ from the oracle"
            L.pgpu_debug_set_wave_decrypt(2)
            assert form() == (5, 64, {1024: 38, 2048: 72, 3072: 112}[bits])
            c = R.op(L.pgpu_batch_encrypt, pk._h, hm, hr, rb)
            assert L.pgpu_batch_row_limbs(c) == 2 * {1024: 38, 2048: 72, 3072: 112}[bits]
            assert R.down(c) == want, "the wavefront-wide kernel differs from the oracle"
            if count <= 300:
                _capi.check(L.pgpu_set_table_gather_policy(1))
                assert R.down(R.op(L.pgpu_batch_encrypt, pk._h, hm, hr, rb)) == want, "masked access differs"
                _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_wave_decrypt(0)                       # its rows, read by the multi-lane kernels
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, c)) == m
            assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, c, c)) == [x * x % nsq for x in want]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_wave_decrypt(1)
        assert (form()[0] == 5) == (count <= 1024)
    finally:
        R.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,threads,batch", [("--threads", 2, 200), ("--threads", 3, 64), ("--threads", 4, 700),
                                                ("--threads-mul", 2, 512), ("--threads-mul", 3, 600), ("--threads-mul", 4, 128)])
def test_small_batches_from_several_api_threads(engine, mode, threads, batch):
    """Several host threads with SMALL vectors at the ipcl:: API (the reference's OpenMP tests, test_cryptography.cpp:45-57 /
    test_ops.cpp, at benchmark sizes): their launches are a mix of the latency forms (while wavefronts x (1 + active
    neighbours) fit the SIMDs: capi.cpp wave_neighbours) and the padded multi-lane forms (launch.hpp place_pad), chosen per
    launch from what the other lanes are doing at that instant.  Every thread's round trips (encrypt + decrypt) resp.
    products (CipherText * PlainText, checked as m * e mod n after a decrypt) must hold whatever the mix."""
    import json
    import subprocess
    from pailliercryptolib_amd import build
    exe = build.build_api_bench()
    for _ in range(2):
        r = subprocess.run([exe, mode, str(threads), str(batch), "6"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert rec.get("round_trip_ok", rec.get("products_ok")) is True


@pytest.mark.gpu
def test_first_decrypts_of_four_threads_at_once(engine):
    """Four threads multiply 1024-element vectors and then decrypt their products for the first time, all at the same moment,
    with launches that share CUs (PGPU_PLACE_PAD=0): each decrypt is the launch that grows its stream's window-table
    workspace.  While workspaces grew through hipMallocAsync / hipFreeAsync, 2 % of such runs returned zeros for the tail of
    a batch (profiles/r06_thread_race.txt); they grow through the block arena now.  150 runs of ~0.3 s (the old code failed 15 of
    400 such runs: a regression shows with probability 0.997 per suite run), every product checked as m * e mod n by the bench itself."""
    import json
    import os
    import subprocess
    from pailliercryptolib_amd import build
    exe = build.build_api_bench()
    env = dict(os.environ, PGPU_PLACE_PAD="0")
    for _ in range(150):
        r = subprocess.run([exe, "--threads-mul", "4", "1024", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
        assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["products_ok"] is True, r.stderr[-2000:]
