"""GPU parity sweep straight at the C-ABI: random moduli of arbitrary bit length (not multiples of
64), padded strides, shared / per-element operands, and the error codes of invalid calls."""
import ctypes
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def to_rows(values, words, stride):
    out = np.zeros((len(values), stride), dtype=np.uint64)
    for i, v in enumerate(values):
        for w in range(words):
            out[i, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_rows(a, words):
    return [sum(int(a[i, w]) << (64 * w) for w in range(words)) for i in range(a.shape[0])]


def test_random_shapes_through_the_c_abi(engine):
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    rng = random.Random(20260928)
    for case in range(60):
        mod_bits = rng.choice([rng.randrange(65, 8100), rng.randrange(65, 1200), 64 * rng.randrange(2, 120)])
        mod = rng.getrandbits(mod_bits) | (1 << (mod_bits - 1)) | 1
        mw = (mod_bits + 63) // 64
        count = rng.choice([1, 2, 3, 5, 17, 33])
        exp_bits = rng.choice([0, 1, 2, 7, 31, 64, 65, 130, 300])
        ew = max(1, (exp_bits + 63) // 64) + rng.randrange(0, 2)          # possibly padded exponent rows
        shared_base, shared_exp = rng.random() < 0.2, rng.random() < 0.3
        bstride = 0 if shared_base else mw + rng.randrange(0, 3)           # padded base rows
        estride = 0 if shared_exp else ew + rng.randrange(0, 2)
        nb, ne = (1 if shared_base else count), (1 if shared_exp else count)
        base = [rng.getrandbits(64 * mw) for _ in range(nb)]               # may exceed the modulus
        exp = [rng.getrandbits(exp_bits) if exp_bits else 0 for _ in range(ne)]
        b_arr = to_rows(base, mw, max(bstride, mw))
        e_arr = to_rows(exp, ew, max(estride, ew))
        m_arr = to_rows([mod], mw, mw)
        out = np.zeros((count, mw), dtype=np.uint64)
        rc = L.pgpu_modexp(ptr(b_arr), bstride, ptr(e_arr), estride, ew, exp_bits, ptr(m_arr), mw, ptr(out), count)
        assert rc == 0, (case, L.pgpu_last_error())
        want = [pow(base[0 if shared_base else i] % mod, exp[0 if shared_exp else i], mod) for i in range(count)]
        assert from_rows(out, mw) == want, (case, mod_bits, exp_bits, count, shared_base, shared_exp)
        # modmul on the same modulus (b scalar-broadcast in a third of the cases)
        a = [rng.getrandbits(64 * mw) for _ in range(count)]
        bs = rng.random() < 0.33
        b = [rng.getrandbits(64 * mw) for _ in range(1 if bs else count)]
        out2 = np.zeros((count, mw), dtype=np.uint64)
        rc = L.pgpu_modmul(ptr(to_rows(a, mw, mw)), ptr(to_rows(b, mw, mw)), 0 if bs else mw, ptr(m_arr), mw,
                           ptr(out2), count)
        assert rc == 0, (case, L.pgpu_last_error())
        assert from_rows(out2, mw) == [x * (b[0] if bs else b[i]) % mod for i, x in enumerate(a)]


def test_invalid_calls_return_error_codes(engine):
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    one = np.ones((1, 1), dtype=np.uint64)
    three = np.array([[3]], dtype=np.uint64)
    out = np.zeros((1, 1), dtype=np.uint64)
    assert L.pgpu_modexp(None, 1, ptr(one), 1, 1, 1, ptr(three), 1, ptr(out), 1) == -1          # null batch
    assert L.pgpu_modexp(ptr(one), 1, ptr(one), 1, 1, 65, ptr(three), 1, ptr(out), 1) == -1     # exp_bits > 64*words
    even = np.array([[4]], dtype=np.uint64)
    assert L.pgpu_modexp(ptr(one), 1, ptr(one), 1, 1, 1, ptr(even), 1, ptr(out), 1) == -2       # even modulus
    big = np.ones((1, 200), dtype=np.uint64)
    outb = np.zeros((1, 200), dtype=np.uint64)
    assert L.pgpu_modexp(ptr(big), 200, ptr(one), 1, 1, 1, ptr(big), 200, ptr(outb), 1) == -3   # 12800-bit modulus
    assert L.pgpu_modexp(ptr(one), 1, ptr(one), 1, 1, 1, ptr(three), 1, ptr(out), 0) == 0       # empty batch is a no-op
    assert L.pgpu_modmul(ptr(one), ptr(one), 1, ptr(even), 1, ptr(out), 1) == -2
    h = ctypes.c_void_p()
    assert L.pgpu_pubkey_create(ptr(even), 1, None, ctypes.byref(h)) == -2
    assert L.pgpu_privkey_create(ptr(three), ptr(three), 1, ctypes.byref(h)) == -6              # p == q
    assert L.pgpu_paillier_encrypt(None, ptr(one), 1, 1, ptr(one), 1, 1, 1, ptr(out), 1) == -1  # null key
    assert b"" != L.pgpu_last_error()
    assert L.pgpu_set_fixed_base_window(15) == -1                # (0..14; the table budgets narrow it further)


def test_batch_api_errors_and_strided_upload(engine):
    """pgpu_batch entry points: argument errors come back as status codes (never exit / abort), a padded host stride is
    accepted, batches of different keys or sizes are refused."""
    import json
    import os
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    rng = random.Random(4)
    h = ctypes.c_void_p()

    def up(vals, words, stride=None):
        stride = stride or words
        a = np.zeros((len(vals), stride), dtype=np.uint64)
        a[:, :words] = ints_to_limbs(vals, words)
        a[:, words:] = np.uint64(0xDEADBEEF)          # padding must be ignored
        b = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_upload(ptr(a), len(vals), words, stride, ctypes.byref(b)))
        return b

    def down(b):
        out = np.empty((L.pgpu_batch_count(b), L.pgpu_batch_words(b)), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(b, ptr(out)))
        return limbs_to_ints(out)
    m = [rng.randrange(n) for _ in range(9)]
    r = [rng.getrandbits(1024) for _ in range(9)]
    bm, br = up(m, 32, stride=40), up(r, 16, stride=17)
    assert down(bm) == m and L.pgpu_batch_words(bm) == 32 and L.pgpu_batch_count(br) == 9
    assert L.pgpu_batch_upload(ptr(np.zeros((2, 2), dtype=np.uint64)), 2, 2, 1, ctypes.byref(h)) == -1    # stride < words
    assert L.pgpu_batch_upload(None, 2, 2, 2, ctypes.byref(h)) == -1
    assert L.pgpu_batch_create(0, 4, ctypes.byref(h)) == -1                                             # empty batch
    assert L.pgpu_batch_encrypt(pk._h, bm, up(r[:5], 16), 1024, ctypes.byref(h)) == -1                   # size mismatch
    c = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_encrypt(pk._h, bm, br, 1024, ctypes.byref(c)))
    assert L.pgpu_batch_is_montgomery(c) == 1 and L.pgpu_batch_is_montgomery(bm) == 0
    assert L.pgpu_batch_encrypt(pk._h, c, br, 1024, ctypes.byref(h)) == -1                               # Montgomery-form "plaintext"
    assert L.pgpu_batch_decrypt_crt(sk._h, bm, ctypes.byref(h)) == -1                                    # wrong width
    assert L.pgpu_batch_ct_add(pk._h, c, up(m[:4], 64), ctypes.byref(h)) == -1                           # size mismatch
    assert L.pgpu_batch_ct_mul(pk._h, c, c, 32, ctypes.byref(h)) == -1                                   # exponents in Montgomery form
    assert L.pgpu_batch_ct_mul(pk._h, c, up([3] * 9, 1), 65, ctypes.byref(h)) == -1                      # e_bits > 64 * words
    # a ciphertext batch of ANOTHER key size is refused by decrypt; one of an equal modulus (a second key object) is fine
    pk2 = engine.PublicKey(n, 2048, hs=hs)
    s = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_ct_add(pk2._h, c, c, ctypes.byref(s)))
    d = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_decrypt_crt(sk._h, s, ctypes.byref(d)))
    assert down(d) == [2 * x % n for x in m]
    assert L.pgpu_batch_download(None, None) == -1
    for b in (bm, br, c, s, d):
        L.pgpu_batch_destroy(b)
    L.pgpu_batch_destroy(None)
