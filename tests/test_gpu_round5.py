"""Round 5 additions, all through the C-ABI against the oracle (reference files cited per test):
  * the one-lane product-scanning CRT-decrypt kernel (csrc/hensel_ps.hpp: 2048-bit keys, 38 limbs of 28 bits per half):
    bit-identical with the default kernels and the oracle -- the two half-width exponentiations of
    PrivateKey::decryptCRT, ipcl/pri_key.cpp:114-146."""
import ctypes
import random

import pytest

from test_gpu_round4 import Res, key_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count", [1, 63, 65, 300, 4100])
def test_ps_decrypt_kernel_is_bit_identical(engine, count):
    """csrc/hensel_ps.hpp: a whole half-width exponentiation per lane by product scanning (2048-bit keys; by default from
    32768 ciphertexts up or beside busy neighbour lanes, forced here): same plaintexts as the default kernels and the
    oracle, ragged batches (padding lanes of the last wavefront), resident ciphertexts from every producer (encrypt,
    CT+CT, CT x PT, uploaded words -- relaxed limbs of the multi-lane kernels and canonical ones), edge plaintexts, and with
    the masked table gather."""
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    p, q, hs = key_case(2048)
    n = p * q
    rng = random.Random(count)
    m = ([0, 1, n - 1] + [rng.randrange(n) for _ in range(count)])[:count]
    m2 = [rng.randrange(n) for _ in range(count)]
    r = [rng.getrandbits(1024) for _ in range(count)]
    e = [rng.getrandbits(33) for _ in range(count)]
    pk, sk = engine.PublicKey(n, 2048, hs=hs), engine.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    R = Res()
    L = R.L
    try:
        c1 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m, 32), R.up(r, 16), 1024)
        c2 = R.op(L.pgpu_batch_encrypt, pk._h, R.up(m2, 32), R.up(r, 16), 1024)
        s = R.op(L.pgpu_batch_ct_add, pk._h, c1, c2)
        t = R.op(L.pgpu_batch_ct_mul, pk._h, s, R.up(e, 1), 33)
        raw = [rng.randrange(1, n * n) for _ in range(count)]        # not encryptions: L_p(c^(p-1)) has no structure
        raw[0] = n * n - 1
        up = R.up(raw, 64)
        up_pair = R.op(L.pgpu_batch_ct_add, pk._h, up, R.up([1], 64))   # the same values as pair rows
        want = [R.down(R.op(L.pgpu_batch_decrypt_crt, sk._h, x)) for x in (c1, s, t, up_pair)]
        assert want[0] == m and want[1] == [(a + b) % n for a, b in zip(m, m2)]
        assert want[2] == [((a + b) * x) % n for a, b, x in zip(m, m2, e)]
        idx = sorted({0, count // 2, count - 1})
        assert [want[3][i] for i in idx] == osk.decrypt([raw[i] for i in idx])
        L.pgpu_debug_set_ps_decrypt(2)
        try:
            split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _capi.check(L.pgpu_decrypt_kernel_form(sk._h, count, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
            assert (split.value, lanes.value, limbs.value) == (4, 1, 38)
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
