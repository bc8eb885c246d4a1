"""GPU parity: the HIP modexp / modmul kernels behind the C-ABI vs the oracle (CPython pow),
bit-exact, on seeded inputs, for every kernel geometry."""
import random

import pytest

from oracle import paillier_oracle as orc

pytestmark = pytest.mark.gpu


def rand_odd(rng, bits):
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


@pytest.mark.parametrize("mod_bits,exp_bits,count", [
    (512, 512, 9), (1024, 512, 8), (1024, 1024, 33), (1536, 768, 5), (2048, 1024, 17),
    (2048, 2048, 8), (3072, 1536, 9), (4096, 1024, 13), (4096, 2048, 4), (6144, 1536, 5),
    (8192, 512, 3), (4096, 32, 9), (4096, 64, 7), (2048, 5, 6), (1000, 333, 7), (4000, 77, 5),
])
def test_modexp_random(engine, mod_bits, exp_bits, count):
    rng = random.Random(mod_bits * 7919 + exp_bits)
    mod = rand_odd(rng, mod_bits)
    base = [rng.randrange(mod) for _ in range(count)]
    exp = [rng.getrandbits(exp_bits) for _ in range(count)]
    got = engine.mod_exp(base, exp, mod)
    want = orc.mod_exp_batch(base, exp, [mod] * count)
    assert got == want


def test_modexp_edge_cases(engine):
    rng = random.Random(99)
    mod = rand_odd(rng, 4096)
    W = 64
    base = [0, 1, mod - 1, mod, mod + 5, (1 << 4096) - 1, 2, rng.randrange(mod), rng.randrange(mod)]
    exp = [5, 0, 3, 7, 1, 2, 0, 1, (1 << 1024) - 1]
    got = engine.mod_exp(base, exp, mod)
    want = [pow(b % mod, e, mod) for b, e in zip(base, exp)]
    assert got == want
    # all-ones limbs, tiny modulus values padded to full width
    for m in (3, 5, (1 << 64) - 1, (1 << 4095) + 1):
        Wm = max(1, (m.bit_length() + 63) // 64)
        b = [rng.getrandbits(64 * Wm) for _ in range(5)]
        e = [rng.getrandbits(70) for _ in range(5)]
        assert engine.mod_exp(b, e, m) == [pow(x % m, y, m) for x, y in zip(b, e)]


@pytest.mark.parametrize("count", [1, 7, 8, 9, 64, 65, 2100])
def test_modexp_batch_sizes_shared_exponent(engine, count):
    """batch sizes the reference's benchmarks use incl. non-multiples of 8
    (bench_cryptography.cpp:19); exponent shared (stride 0), as in decryptCRT."""
    rng = random.Random(count)
    mod = rand_odd(rng, 2048)
    base = [rng.randrange(mod) for _ in range(count)]
    e = rng.getrandbits(1024 if count < 100 else 64)
    got = engine.mod_exp(base, [e], mod)
    assert got == [pow(b, e, mod) for b in base]


def test_modexp_shared_base(engine):
    rng = random.Random(5)
    mod = rand_odd(rng, 4096)
    hs = rng.randrange(mod)
    r = [rng.getrandbits(1024) for _ in range(10)]
    assert engine.mod_exp([hs], r, mod) == [pow(hs, x, mod) for x in r]


@pytest.mark.parametrize("mod_bits,count", [(1024, 9), (2048, 33), (4096, 70), (6144, 5)])
def test_modmul(engine, mod_bits, count):
    rng = random.Random(mod_bits + count)
    mod = rand_odd(rng, mod_bits)
    a = [rng.randrange(mod) for _ in range(count)]
    b = [rng.randrange(mod) for _ in range(count)]
    assert engine.mod_mul(a, b, mod) == orc.ct_add(a, b, mod)
    assert engine.mod_mul(a, b[:1], mod) == orc.ct_add(a, b[:1], mod)   # scalar broadcast
    # operands >= modulus are reduced
    big = [(1 << mod_bits) - 1 - i for i in range(count)]
    assert engine.mod_mul(big, big, mod) == [x * x % mod for x in big]


def test_errors(engine):
    import pailliercryptolib_amd as pa
    with pytest.raises(pa._capi.PgpuError):
        engine.mod_exp([3], [5], 1 << 64)          # even modulus
    with pytest.raises(RuntimeError):
        engine.mod_exp([3, 4, 5], [5, 6], 7)       # size mismatch


@pytest.mark.parametrize("mod_bits,exp_bits,count", [(1024, 1024, 256), (2048, 1024, 512), (4096, 2048, 128)])
def test_modexp_vs_openssl_bulk(engine, mod_bits, exp_bits, count):
    """The reference's accelerator tests check batched modexp against OpenSSL BN_mod_exp
    (module/heqat/test/test_bnModExp.cpp:57-60,205-208); same check for the HIP path on bulk
    batches that CPython pow would be slow for."""
    import numpy as np
    from oracle import c_oracle
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    if c_oracle.openssl_lib() is None:
        pytest.skip("libcrypto not available at build time")
    rng = random.Random(mod_bits + count)
    mod = rand_odd(rng, mod_bits)
    W, E = mod_bits // 64, exp_bits // 64
    nrng = np.random.default_rng(mod_bits)
    base_l = nrng.integers(0, 1 << 63, size=(count, W), dtype=np.uint64) << np.uint64(1)
    base_l[:, W - 1] >>= np.uint64(2)                       # < 2^(bits-1) <= mod
    exp_l = nrng.integers(0, 1 << 63, size=(count, E), dtype=np.uint64)
    want = c_oracle.openssl_modexp_batch(base_l, exp_l, ints_to_limbs([mod], W)[0])
    got = engine.mod_exp(limbs_to_ints(base_l), limbs_to_ints(exp_l), mod)
    assert got == limbs_to_ints(want)


def test_modexp_shared_exponent_schedules(engine):
    """One exponent for the whole batch (>= 16 elements) runs a host-built sliding-window schedule instead of
    the per-element digit scan: exercise its corner cases -- single bit, all ones, long zero runs (squarings
    only, also more than 1023 in a row), trailing zeros, tiny exponents -- at two modulus widths."""
    rng = random.Random(4242)
    for mod_bits in (1024, 4096):
        mod = rand_odd(rng, mod_bits)
        base = [rng.randrange(mod) for _ in range(24)]
        exps = [1, 2, 3, 5, 64, (1 << 300), (1 << 300) - 1, (1 << 1100) + 1, ((1 << 1100) + 1) << 37,
                0x5555555555555555555555555555, int("1" + "0" * 70 + "1" * 9 + "0" * 33 + "101", 2),
                rng.getrandbits(1024) | (1 << 1023), rng.getrandbits(2048) << 5]
        for e in exps:
            got = engine.mod_exp(base, [e], mod)
            assert got == [pow(b, e, mod) for b in base], (mod_bits, hex(e)[:40])


@pytest.mark.parametrize("mod_bits", [1024, 2048, 3072, 4096, 6144])
def test_modexp_saturated_limbs(engine, mod_bits):
    """Column-accumulator headroom: moduli 2^bits - c (every limb all ones), operands mod-1, mod-2, all-ones
    bit patterns and an all-ones exponent keep every 29-bit limb, every quotient digit and every carry at its
    maximum through whole exponentiations (the relaxed-limb epilogue is sized for exactly this case)."""
    rng = random.Random(mod_bits)
    for c in (1, 3, 189):
        mod = (1 << mod_bits) - c
        base = [mod - 1, mod - 2, (1 << (mod_bits - 1)) - 1, mod >> 1, int("1" * (mod_bits - 3), 2)] * 4
        base += [rng.randrange(mod) for _ in range(4)]
        for e in ((1 << 200) - 1, (1 << 64) | 1):
            got = engine.mod_exp(base, [e] * len(base), mod)
            assert got == [pow(b, e, mod) for b in base], (mod_bits, c)
        got = engine.mod_mul(base, base[::-1], mod)
        assert got == [(x * y) % mod for x, y in zip(base, base[::-1])]


@pytest.mark.parametrize("mod_bits,count", [(2048, 5), (2048, 130), (2040, 33), (3072, 7), (3072, 70)])
def test_modexp_small_batches_latency_split(engine, mod_bits, count):
    """Batches that leave SIMDs idle run 16 lanes per element (Geo<16,5> / Geo<16,7>, capi.hip: latency_geo),
    with their own Montgomery context where the limb count differs; same bits as every other split."""
    rng = random.Random(mod_bits + count)
    mod = rand_odd(rng, mod_bits)
    base = [rng.randrange(mod) for _ in range(count)]
    exp = [rng.getrandbits(300) for _ in range(count)]
    assert engine.mod_exp(base, exp, mod) == [pow(b, e, mod) for b, e in zip(base, exp)]
    e = rng.getrandbits(1024)
    assert engine.mod_exp(base, [e], mod) == [pow(b, e, mod) for b in base]


@pytest.mark.parametrize("row_source", [0, 1])
def test_both_kernel_forms_agree_with_pow(engine, row_source):
    """modexp_kernel exists in two forms per geometry -- multiplier rows read from LDS, or broadcast from registers
    (the form for launches that leave a wavefront alone on its SIMD) -- normally picked by the wavefront count.
    Force each form over every geometry class, several batch sizes, squarings and table products, fixed-window scan
    and (shared exponent through the host API) sliding schedule; everything against CPython pow."""
    import ctypes
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    L.pgpu_debug_set_row_source.argtypes = [ctypes.c_int]
    L.pgpu_debug_set_row_source.restype = None
    rng = random.Random(900 + row_source)
    L.pgpu_debug_set_row_source(row_source)
    try:
        for bits, ebits, count in ((512, 70, 33), (1024, 300, 200), (1536, 129, 70), (2048, 1024, 40), (2048, 64, 300),
                                   (3072, 200, 24), (4096, 96, 40)):
            mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
            base = [rng.randrange(mod) for _ in range(count)]
            exp = [rng.getrandbits(ebits) for _ in range(count)]
            assert engine.mod_exp(base, exp, mod) == [pow(b, e, mod) for b, e in zip(base, exp)], (bits, ebits, count)
            e1 = rng.getrandbits(ebits) | 1
            assert engine.mod_exp(base, [e1], mod) == [pow(b, e1, mod) for b in base], (bits, "shared", count)
    finally:
        L.pgpu_debug_set_row_source(-1)


@pytest.mark.parametrize("row_source", [0, 1])
@pytest.mark.parametrize("policy", [0, 1])
def test_both_kernel_forms_in_crt_decrypt(engine, row_source, policy):
    """The two-context decrypt launch (p^2 / q^2 sides, fixed-window scan or sliding schedules with parity waves) in
    both kernel forms, for 1024-, 2048- and 3072-bit keys and batch sizes on either side of the latency-geometry
    switch; ciphertexts come from the oracle, the plaintexts must come back."""
    import ctypes
    import json
    import os
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    L.pgpu_debug_set_row_source.argtypes = [ctypes.c_int]
    L.pgpu_debug_set_row_source.restype = None
    gold = os.path.join(os.path.dirname(__file__), "golden", "seeded_vectors.json")
    rng = random.Random(77 + 2 * row_source + policy)
    old = L.pgpu_get_secret_exponent_policy()
    L.pgpu_debug_set_row_source(row_source)
    _capi.check(L.pgpu_set_secret_exponent_policy(policy))
    try:
        for case in [c for c in json.load(open(gold))["cases"] if c["djn"]]:
            p, q, hs, bits = int(case["p"], 16), int(case["q"], 16), int(case["hs"], 16), case["bits"]
            n = p * q
            opk = orc.PublicKey(n, bits)
            opk.set_djn(hs)
            sk = engine.PrivateKey(p, q)
            for count in (3, 70):
                m = [rng.randrange(n) for _ in range(count)]
                ct = opk.encrypt(m, [rng.getrandbits(bits // 2) for _ in range(count)])
                assert sk.decrypt(ct) == m, (bits, count)
    finally:
        L.pgpu_debug_set_row_source(-1)
        _capi.check(L.pgpu_set_secret_exponent_policy(old))
