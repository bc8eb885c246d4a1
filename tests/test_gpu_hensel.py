"""CRT decrypt through the split-form exponentiation (csrc/hensel.hpp: residues modulo p^2 as pairs a - P*b) against
the full-width kernel and the oracle: same bits for every key class it is compiled for, both exponent policies,
ragged batch sizes, the KAT and the seeded fixtures, ciphertexts that are not encryptions, and Montgomery-form
(device-resident) ciphertexts."""
import ctypes
import json
import os
import random

import pytest

from oracle import paillier_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _lib():
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    L.pgpu_debug_set_hensel.argtypes = [ctypes.c_int]
    L.pgpu_debug_set_hensel.restype = None
    return L


@pytest.fixture()
def hensel(engine):
    L = _lib()
    yield L.pgpu_debug_set_hensel
    L.pgpu_debug_set_hensel(1)   # the library default


def _kat():
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    return {key: (int(v, 16) if isinstance(v, str) and v.startswith("0x") else v) for key, v in k.items()}


@pytest.mark.parametrize("form", [2, 3])
def test_kat_and_fixtures_through_split_form(engine, hensel, form):
    hensel(form)    # 2: the throughput form, 3: the latency form, whatever the batch size
    kat = _kat()
    sk = engine.PrivateKey(kat["p"], kat["q"])
    num = kat["num_values"]
    m = [kat["m0"]] * num
    m[1] = kat["m1"]
    c = [kat["c1"]] * num
    c[1] = kat["c2"]
    assert sk.decrypt(c) == m
    assert sk.decrypt([kat["c1c2"]]) == [kat["m1m2"]]
    data = json.load(open(os.path.join(GOLD, "seeded_vectors.json")))
    ran = 0
    for case in data["cases"]:
        p, q = int(case["p"], 16), int(case["q"], 16)
        sk = engine.PrivateKey(p, q)
        assert sk.decrypt([int(v, 16) for v in case["c"]]) == [int(v, 16) for v in case["m"]], case["bits"]
        ran += 1
    assert ran


@pytest.mark.parametrize("bits,count", [(1024, 37), (1024, 2311), (2048, 1), (2048, 16), (2048, 17), (2048, 2500), (3072, 5), (3072, 2100),
                                        (4096, 3), (4096, 2051), (2037, 9), (2037, 2200)])
@pytest.mark.parametrize("policy", ["fixed", "sliding"])
@pytest.mark.parametrize("form", [2, 3])
def test_split_form_equals_full_width(engine, hensel, bits, count, policy, form):
    """Random elements of Z_{n^2} (mostly NOT encryptions, so L_p(c^(p-1)) has no structure to hide behind)."""
    from pailliercryptolib_amd import _capi
    if bits == 4096 and not _capi.lib().pgpu_build_features() & _capi.FEATURE_4096_SPLIT:
        pytest.skip("the split forms of the 4096-bit key class are not in this build (PGPU_BUILD_4096=1 builds them)")
    rng = random.Random(bits * 7 + count)
    if bits == 2048:
        kat = _kat()
        p, q = kat["p"], kat["q"]
    elif bits in (4096, 2037):     # 2037: a 1013-bit and a 1024-bit prime (five ciphertext chunks per side)
        k4 = json.load(open(os.path.join(GOLD, "primes_4096.json" if bits == 4096 else "primes_uneven.json")))
        p, q = int(k4["p"], 16), int(k4["q"], 16)
    else:
        case = next(c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == bits)
        p, q = int(case["p"], 16), int(case["q"], 16)
    n = p * q
    _capi.check(_capi.lib().pgpu_set_secret_exponent_policy(1 if policy == "sliding" else 0))
    try:
        sk = engine.PrivateKey(p, q)
        c = [rng.randrange(1, n * n) for _ in range(count)]
        c[0] = 1
        if count > 2:
            c[1] = n * n - 1
            c[2] = n + 1
        hensel(0)
        ref = sk.decrypt(c)
        hensel(form)
        got = sk.decrypt(c)
        assert got == ref
        osk = orc.PrivateKey(n, p, q)
        for i in sorted(set([0, 1 % count, 2 % count, count - 1, count // 2])):
            assert got[i] == osk.decrypt([c[i]])[0]
    finally:
        _capi.check(_capi.lib().pgpu_set_secret_exponent_policy(0))


def test_split_form_roundtrip_and_resident_chain(engine, hensel):
    """enc -> dec at a size that takes the split form by default, and the device-resident chain
    encrypt -> CT+CT -> decrypt, whose ciphertexts reach decrypt in the Montgomery form of n^2."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    kat = _kat()
    p, q = kat["p"], kat["q"]
    n = p * q
    nw = 32
    rng = random.Random(11)
    pk = engine.PublicKey(n, 2048, hs=kat["bench_hs"])
    sk = engine.PrivateKey(p, q)
    count = 3000
    m1 = [rng.getrandbits(64) for _ in range(count)]
    m2 = [rng.getrandbits(64) for _ in range(count)]
    r1 = [rng.getrandbits(1024) for _ in range(count)]
    r2 = [rng.getrandbits(1024) for _ in range(count)]
    hensel(1)
    assert sk.decrypt(pk.encrypt(m1, r1)) == m1

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def up(vals, words):
        h = ctypes.c_void_p()
        a = ints_to_limbs(vals, words)
        _capi.check(L.pgpu_batch_upload(ptr(a), len(vals), words, words, ctypes.byref(h)))
        return h

    def op(fn, *a):
        h = ctypes.c_void_p()
        _capi.check(fn(*a, ctypes.byref(h)))
        return h
    hs = [up(m1, nw), up(m2, nw), up(r1, 16), up(r2, 16)]
    c1 = op(L.pgpu_batch_encrypt, pk._h, hs[0], hs[2], 1024)
    c2 = op(L.pgpu_batch_encrypt, pk._h, hs[1], hs[3], 1024)
    assert L.pgpu_batch_is_montgomery(c1)
    sm = op(L.pgpu_batch_ct_add, pk._h, c1, c2)
    d = op(L.pgpu_batch_decrypt_crt, sk._h, sm)
    out = np.empty((L.pgpu_batch_count(d), L.pgpu_batch_words(d)), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(d, ptr(out)))
    assert limbs_to_ints(out) == [a + b for a, b in zip(m1, m2)]
    for h in hs + [c1, c2, sm, d]:
        L.pgpu_batch_destroy(h)


@pytest.mark.parametrize("fbw", [4, 12])
def test_split_form_djn_encrypt_equals_full_width(engine, hensel, fbw):
    """DJN encrypt through the split-form fixed-base kernel (hensel_fb_encrypt_kernel) against the full-width
    fb_encrypt_kernel and the oracle: edge plaintexts (0, 1, n-1, >= n up to the row width), edge randomness
    (0, 1, short, full width), plain and Montgomery-form (device-resident) results."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    kat = _kat()
    p, q = kat["p"], kat["q"]
    n = p * q
    nw = 32
    assert n.bit_length() == 64 * nw                         # the split form needs plaintext rows no wider than n
    rng = random.Random(fbw)
    count = 203
    m = [0, 1, n - 1, n, n + 5, (1 << 2048) - 1] + [rng.randrange(n) for _ in range(count - 6)]
    r = [0, 1, 3, (1 << 1024) - 1, 1 << 1023, rng.getrandbits(17)] + [rng.getrandbits(1024) for _ in range(count - 6)]
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(kat["bench_hs"])
    want = opk.encrypt(m, r)
    _capi.check(L.pgpu_set_fixed_base_window(fbw))
    try:
        pk = engine.PublicKey(n, 2048, hs=kat["bench_hs"])
        split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        hensel(1)      # by batch size: the split form from 8192 elements per device up
        _capi.check(L.pgpu_encrypt_kernel_form(pk._h, nw, 8192, ctypes.byref(split), ctypes.byref(lanes),
                                               ctypes.byref(limbs)))
        assert (split.value, lanes.value, limbs.value) == (1, 8, 18)
        _capi.check(L.pgpu_encrypt_kernel_form(pk._h, nw, count, ctypes.byref(split), ctypes.byref(lanes),
                                               ctypes.byref(limbs)))
        assert split.value == 0
        hensel(2)      # forced for this small batch
        _capi.check(L.pgpu_encrypt_kernel_form(pk._h, nw, count, ctypes.byref(split), ctypes.byref(lanes),
                                               ctypes.byref(limbs)))
        assert split.value == 1
        got = pk.encrypt(m, r)
        assert got == want
        hensel(0)
        _capi.check(L.pgpu_encrypt_kernel_form(pk._h, nw, count, ctypes.byref(split), ctypes.byref(lanes),
                                               ctypes.byref(limbs)))
        assert split.value == 0
        assert pk.encrypt(m, r) == want
        # Montgomery-form result of a resident batch, downloaded (leaves the domain) and decrypted in place
        hensel(2)

        def ptr(a):
            return a.ctypes.data_as(ctypes.c_void_p)

        def up(vals, words):
            h = ctypes.c_void_p()
            a = ints_to_limbs(vals, words)
            _capi.check(L.pgpu_batch_upload(ptr(a), len(vals), words, words, ctypes.byref(h)))
            return h
        bm, br, c = up(m, nw), up(r, 16), ctypes.c_void_p()
        _capi.check(L.pgpu_batch_encrypt(pk._h, bm, br, 1024, ctypes.byref(c)))
        assert L.pgpu_batch_is_montgomery(c)
        out = np.empty((count, 2 * nw), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(c, ptr(out)))
        assert limbs_to_ints(out) == want
        sk = engine.PrivateKey(p, q)
        d = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))
        dm = np.empty((count, nw), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(d, ptr(dm)))
        assert limbs_to_ints(dm) == [v % n for v in m]
        for h in (bm, br, c, d):
            L.pgpu_batch_destroy(h)
    finally:
        _capi.check(L.pgpu_set_fixed_base_window(13))


@pytest.mark.parametrize("count", [37, 4200])
def test_split_form_ct_mul_and_plain_obfuscator_equal_full_width(engine, hensel, count):
    """The generic split-form exponentiation modulo n^2 (hensel_modexp_kernel: 8 lanes per half for small batches, 4
    beyond 4096 elements) against the full-width modexp_kernel and the oracle: CT x PT with per-element exponents of
    every width on plain and on Montgomery-form (resident) ciphertexts, a broadcast scalar exponent, and the non-DJN
    encrypt r^n * (1 + n*m) with edge plaintexts."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    kat = _kat()
    p, q = kat["p"], kat["q"]
    n = p * q
    nsq = n * n
    nw = 32
    rng = random.Random(count)
    c = [rng.randrange(1, nsq) for _ in range(count)]
    c[0], c[1] = 1, nsq - 1
    widths = [0, 1, 2, 31, 32, 33, 64, 100, 2048]
    e = [rng.getrandbits(widths[i % len(widths)]) for i in range(count)]
    m = [0, 1, n - 1, n + 7, (1 << 2048) - 1] + [rng.randrange(n) for _ in range(count - 5)]
    r = [1, 2, n - 1] + [rng.randrange(1, n) for _ in range(count - 3)]
    # the oracle on a sample (all of a small batch); the rest of a large batch is compared between the two kernels
    idx = list(range(count)) if count <= 64 else list(range(48)) + list(range(count - 16, count))
    want_mul = {i: pow(c[i], e[i], nsq) for i in idx}
    want_scalar = {i: pow(c[i], 65537, nsq) for i in idx}
    opk = orc.PublicKey(n, 2048)
    want_enc = {i: opk.encrypt([m[i]], [r[i]])[0] for i in idx}
    seen = {}

    def check(name, got, want, mode):
        assert all(got[i] == want[i] for i in idx), (name, mode)
        assert seen.setdefault(name, got) == got, (name, "split form and full-width kernel differ")

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

    pk = engine.PublicKey(n, 2048)                      # non-DJN: r^n
    for mode in (1, 0):
        hensel(mode)
        check("enc", pk.encrypt(m, r), want_enc, mode)
        bc, be, bs = up(c, 2 * nw), up(e, nw), up([65537], 1)
        t = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_ct_mul(pk._h, bc, be, 2048, ctypes.byref(t)))       # plain ciphertexts in
        assert L.pgpu_batch_is_montgomery(t)
        got_mul = down(t)
        check("mul", got_mul, want_mul, mode)
        u = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_ct_mul(pk._h, t, bs, 17, ctypes.byref(u)))          # Montgomery form in, scalar
        check("mul2", down(u), {i: pow(want_mul[i], 65537, nsq) for i in idx}, mode)
        v = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_ct_mul(pk._h, bc, bs, 17, ctypes.byref(v)))
        check("scalar", down(v), want_scalar, mode)
        for h in (bc, be, bs, t, u, v):
            L.pgpu_batch_destroy(h)


@pytest.mark.parametrize("pbits", [256, 384, 640, 768, 1280])
def test_split_form_odd_key_sizes(engine, hensel, pbits):
    """Keys between the standard classes take the smallest compiled form with enough limbs (a 1536-bit key runs the
    2048-bit forms): CRT decrypt through every form against the full-width kernel and the oracle."""
    import sys
    sys.path.insert(0, GOLD)
    import gen_primes
    rng = random.Random(pbits)
    p, q = gen_primes.prime(pbits, rng, top2=False), gen_primes.prime(pbits - 3, rng, top2=False)
    n = p * q
    sk = engine.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    for count in (6, 2100):
        c = [rng.randrange(1, n * n) for _ in range(count)]
        hensel(0)
        ref = sk.decrypt(c)
        for form in (1, 2, 3):
            hensel(form)
            assert sk.decrypt(c) == ref, (pbits, count, form)
        for i in (0, count - 1):
            assert ref[i] == osk.decrypt([c[i]])[0]


@pytest.mark.parametrize("count", [5, 4300])
def test_perfect_square_moduli_through_the_generic_seam(engine, hensel, count):
    """pgpu_modexp (the ipcl::modExp seam) detects a modulus that is the square of an odd root and runs the split form
    for it: n^2 and p^2 of the KAT key, an odd root size in between, bases above the modulus, exponent 0 / 1 / wide,
    a shared (scalar) exponent; bit-identical to the full-width kernel and to pow.  A modulus that only looks like
    a square modulo 16 takes the full-width kernel as before."""
    kat = _kat()
    p, q = kat["p"], kat["q"]
    n = p * q
    rng = random.Random(count)
    root3 = rng.getrandbits(1500) | (1 << 1499) | 1
    for mod in (n * n, p * p, root3 * root3, n * n + 16):
        bits = mod.bit_length()
        base = [rng.randrange(mod) for _ in range(count)]
        base[0] = 0
        base[1] = mod + 5 if (mod + 5).bit_length() <= 64 * ((bits + 63) // 64) else mod - 1
        exp = [rng.getrandbits([0, 1, 32, 64, 300][i % 5]) for i in range(count)]
        idx = list(range(count)) if count <= 64 else list(range(24)) + list(range(count - 8, count))
        hensel(0)
        ref = engine.mod_exp(base, exp, mod)
        ref_s = engine.mod_exp(base, [65537] * count, mod)
        hensel(1)
        assert engine.mod_exp(base, exp, mod) == ref, bits
        assert engine.mod_exp(base, [65537] * count, mod) == ref_s, bits
        for i in idx:
            assert ref[i] == pow(base[i], exp[i], mod) and ref_s[i] == pow(base[i], 65537, mod), (bits, i)


@pytest.mark.parametrize("m_words", [1, 3])
def test_split_form_encrypt_narrow_plaintext_rows(engine, hensel, m_words):
    """Plaintext rows narrower than n (the ipcl:: layer packs u32 / u64 plaintexts into one word) at a batch that takes
    the split-form fixed-base kernel by size: three ciphertexts against the oracle, all through the round trip."""
    import numpy as np
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    L = _capi.lib()
    kat = _kat()
    p, q = kat["p"], kat["q"]
    n = p * q
    pk, sk = engine.PublicKey(n, 2048, hs=kat["bench_hs"]), engine.PrivateKey(p, q)
    opk = orc.PublicKey(n, 2048)
    opk.set_djn(kat["bench_hs"])
    rng = random.Random(m_words)
    count = 8200
    m = [rng.getrandbits(64 * m_words - 1) for _ in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    hensel(1)
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_encrypt_kernel_form(pk._h, m_words, count, ctypes.byref(split), ctypes.byref(lanes),
                                           ctypes.byref(limbs)))
    assert split.value == 1

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)
    hm, hr, c, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    am, ar = ints_to_limbs(m, m_words), ints_to_limbs(r, 16)
    _capi.check(L.pgpu_batch_upload(ptr(am), count, m_words, m_words, ctypes.byref(hm)))
    _capi.check(L.pgpu_batch_upload(ptr(ar), count, 16, 16, ctypes.byref(hr)))
    _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, 1024, ctypes.byref(c)))
    out = np.empty((count, 64), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(c, ptr(out)))
    ct = limbs_to_ints(out)
    for i in (0, 1, count - 1):
        assert ct[i] == opk.encrypt([m[i]], [r[i]])[0]
    _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))
    dm = np.empty((count, 32), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(d, ptr(dm)))
    assert limbs_to_ints(dm) == m
    for h in (hm, hr, c, d):
        L.pgpu_batch_destroy(h)
