"""Round 6, through the C-ABI against the oracle:
  * the one-lane product-scanning kernel of the n^2 domain (csrc/hensel_ps_n2.hpp: 2048-bit keys, 75 limbs of 28 bits per
    half, a whole exponentiation per lane) -- CipherText * PlainText on resident batches, ipcl/ciphertext.cpp:143-162:
    bit-identical with pow(c, e, n^2) and with the multi-lane kernels, for every producer of resident ciphertexts, ragged
    batches, edge exponents and bases, per-element and broadcast exponents, 33- to 1024-bit plaintexts (the reference's
    tests use 32-bit ones, test_ops.cpp:294-325; its benchmark 1024-bit ones, bench_ops.cpp:138-149), masked table access."""
import ctypes
import random

import pytest

from test_gpu_round4 import Res, key_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count,e_bits", [(1, 33), (63, 33), (65, 64), (300, 33), (300, 1024), (4100, 17), (4100, 40)])
def test_ps_modexp_n2_kernel_is_bit_identical(engine, count, e_bits):
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048)
    n = p * q
    nsq = n * n
    nw, ew = 32, (e_bits + 63) // 64
    rng = random.Random(count * 1000 + e_bits)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    r = [rng.getrandbits(1024) for _ in range(count)]
    e = ([0, 1, (1 << e_bits) - 1, 2] + [rng.getrandbits(e_bits) for _ in range(count)])[:count]
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    R = Res()
    L = R.L
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, nw), R.up(r, nw // 2), 1024)      # pair rows from the encrypt kernel
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c1)                                     # ... from CT + CT (relaxed limbs)
        raw = ([0, 1, nsq - 1, n, n + 1] + [rng.randrange(nsq) for _ in range(count)])[:count]   # arbitrary residues, incl. 0 and multiples of n
        up = R.up(raw, 2 * nw)                                                            # ... uploaded words (converted on the way in)
        eh = R.up(e, ew)
        srcs = [c1, s, up]
        vals = [R.down(x) for x in srcs]
        want = [[pow(c, x, nsq) for c, x in zip(v, e)] for v in vals]
        base = [R.down(R.op(L.pgpu_batch_ct_mul, pk._h, x, eh, e_bits)) for x in srcs]   # the default kernels at this size
        assert base == want, "the multi-lane kernels differ from pow()"
        L.pgpu_debug_set_ps_decrypt(2)
        try:
            split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _capi.check(L.pgpu_modexp_n2_kernel_form(pk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
            assert (split.value, lanes.value, limbs.value) == (4, 1, 75)
            outs = [R.op(L.pgpu_batch_ct_mul, pk._h, x, eh, e_bits) for x in srcs]
            assert [R.down(o) for o in outs] == want, "the one-lane kernel differs from pow()"
            assert all(L.pgpu_batch_row_limbs(o) == 144 for o in outs)                   # results stay pair rows ...
            t2 = R.op(L.pgpu_batch_ct_mul, pk._h, outs[0], eh, e_bits)                    # ... that every kernel reads again:
            assert R.down(t2) == [pow(c, x, nsq) for c, x in zip(want[0], e)]             # itself,
            L.pgpu_debug_set_ps_decrypt(1)
            assert R.down(R.op(L.pgpu_batch_ct_add, pk._h, outs[0], c1)) == [a * b % nsq for a, b in zip(want[0], vals[0])]   # CT + CT,
            assert R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, outs[0])) == [a * x % n for a, x in zip(m, e)]               # CRT decrypt
            L.pgpu_debug_set_ps_decrypt(2)
            one_e = R.up([e[-1]], ew)                                                     # one exponent for the whole batch
            assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, c1, one_e, e_bits)) == [pow(c, e[-1], nsq) for c in vals[0]]
            if count <= 300:
                _capi.check(L.pgpu_set_table_gather_policy(1))                            # every table entry read, one selected
                assert R.down(R.op(L.pgpu_batch_ct_mul, pk._h, s, eh, e_bits)) == want[1]
        finally:
            _capi.check(L.pgpu_set_table_gather_policy(0))
            L.pgpu_debug_set_ps_decrypt(1)
    finally:
        R.close()


def test_ps_modexp_n2_kernel_by_size(engine):
    """65536 resident ciphertexts put a wavefront on every SIMD at 64 exponentiations per wavefront: the default policy takes
    the one-lane kernel (csrc/policy.cpp: modexp_ps_form_pays); 65536 + 4096 would waste most of a second round and keep
    the multi-lane form.  Sampled against pow(), and the whole output against the multi-lane kernel's."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import limbs_to_ints
    p, q, hs = key_case(2048)
    n = p * q
    nsq = n * n
    pk = engine.PublicKey(n, 2048, hs=hs)
    L = _capi.lib()
    count, W = 65536, 64
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_modexp_n2_kernel_form(pk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    assert (split.value, lanes.value, limbs.value) == (4, 1, 75)
    _capi.check(L.pgpu_modexp_n2_kernel_form(pk._h, count + 4096, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    assert split.value == 2
    rng = np.random.default_rng(66)
    a = np.frombuffer(rng.bytes(count * W * 8), dtype=np.uint64).reshape(count, W).copy()
    a[:, -1] &= np.uint64((1 << 60) - 1)
    e = np.frombuffer(rng.bytes(count * 8), dtype=np.uint64).reshape(count, 1).copy() & np.uint64((1 << 32) - 1)
    ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    live = []

    def op(fn, *args):
        h = ctypes.c_void_p()
        _capi.check(fn(*args, ctypes.byref(h)))
        live.append(h)
        return h
    try:
        ha = op(L.pgpu_batch_upload, ptr(a), count, W, W)
        he = op(L.pgpu_batch_upload, ptr(e), count, 1, 1)
        got = np.empty((count, W), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(op(L.pgpu_batch_ct_mul, pk._h, ha, he, 32), ptr(got)))
        idx = [0, 1, 63, 64, 32767, 32768, 65535]
        assert limbs_to_ints(got[idx]) == [pow(b, int(x), nsq) for b, x in zip(limbs_to_ints(a[idx]), e[idx, 0])]
        L.pgpu_debug_set_ps_decrypt(0)
        try:
            ref = np.empty((count, W), dtype=np.uint64)
            _capi.check(L.pgpu_batch_download(op(L.pgpu_batch_ct_mul, pk._h, ha, he, 32), ptr(ref)))
        finally:
            L.pgpu_debug_set_ps_decrypt(1)
        assert np.array_equal(got, ref), "one-lane and multi-lane CT x PT differ"
    finally:
        for h in live:
            L.pgpu_batch_destroy(h)
