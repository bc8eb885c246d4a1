"""Integer model of the split-form arithmetic of csrc/hensel.hpp (CPU test, no GPU): residues modulo P^2 as pairs
x == a - P*b, the pair Montgomery product built from two half-width reductions, the chunked entry of a
ciphertext, the fixed-window exponentiation, and the exit under the true prime that yields
mp = L_p(c^(p-1) mod p^2) * hp mod p (ipcl/pri_key.cpp:136-157) without a division.  The lazy bounds the kernel
relies on (every component below 2P) are asserted on the way."""
import json
import os
import random

import pytest

LB = 29
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def setup(p, K):
    k = (-pow(p, -1, 1 << LB)) % (1 << LB)
    P = k * p
    assert P % (1 << LB) == (1 << LB) - 1            # unit quotient digits
    R = 1 << (LB * 2 * K)
    assert R >= 256 * P
    return k, P, R


def to_pair(z, P):
    z %= P * P
    a, f = z % P, z // P
    return (a, (P - f) % P)


def val(pr, P):
    return (pr[0] - P * pr[1]) % (P * P)


def redc(T, P, R):
    q = (T * (-pow(P, -1, R))) % R
    return (T + q * P) // R, q


def pmul(x, y, P, R):
    (a, b), (c, d) = x, y
    t, q = redc(a * c, P, R)
    w, _ = redc(a * d + b * c + q, P, R)
    assert t < 2 * P and w < 2 * P
    return (t, w)


def cases():
    out = []
    for case in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"]:
        if case["bits"] in (1024, 2048):
            out.append((int(case["p"], 16), int(case["q"], 16)))
    return out[:2]


@pytest.mark.parametrize("side", [0, 1])
def test_pair_exponentiation_matches_reference_formula(side):
    rng = random.Random(3 + side)
    for p, q in cases():
        if side:
            p, q = q, p
        bits = p.bit_length()
        K = (bits + 37 + 2 * LB - 1) // (2 * LB)
        k, P, R = setup(p, K)
        n = p * q
        cw = min((bits + 63) // 64, bits // 64)
        nch = (2 * ((n.bit_length() + 63) // 64) + cw - 1) // cw
        conv = [to_pair((1 << (64 * cw * i)) * R * R, P) for i in range(nch)]
        hp = pow(((pow(n + 1, p - 1, p * p) - 1) // p), -1, p)        # computeHfun, pri_key.cpp:159-167
        for c in (1, n + 1, n * n - 1, rng.randrange(n * n), rng.randrange(n * n)):
            acc = (0, 0)
            for i in range(nch):
                z = (c >> (64 * cw * i)) & ((1 << (64 * cw)) - 1)
                assert z < 2 * P
                t = pmul((z, 0), conv[i], P, R)
                acc = (acc[0] + t[0], acc[1] + t[1])
            assert val(acc, P) == c * R % (P * P)
            w = 5
            tbl = [to_pair(R, P), acc]
            for _ in range(2, 1 << w):
                tbl.append(pmul(tbl[-1], acc, P, R))
            e = p - 1
            nwin = (e.bit_length() + w - 1) // w
            x = tbl[(e >> (w * (nwin - 1))) & 31]
            for i in range(nwin - 2, -1, -1):
                for _ in range(w):
                    x = pmul(x, x, P, R)
                x = pmul(x, tbl[(e >> (w * i)) & 31], P, R)
            u = pow(c, p - 1, p * p)
            assert val(x, P) % (p * p) == u * R % (p * p)
            # exit: (a, k*b mod p) is a pair modulo p^2; product with (hp, 0) under the true prime
            a, B = x[0], redc(x[1] * (k * R % p), p, R)[0]            # k*b mod p (lazy) by a half-width product
            assert B < 2 * p and (B - k * x[1]) % p == 0
            n0 = (-pow(p, -1, R)) % R
            q1 = a * hp * n0 % R
            t = (a * hp + q1 * p) // R
            q2 = (B * hp + q1) * n0 % R
            w2 = (B * hp + q1 + q2 * p) // R
            assert t < 2 * p and w2 < 2 * p and t % p == hp
            j = 1 if t >= p else 0
            assert (j - w2) % p == ((u - 1) // p) * hp % p          # pri_key.cpp:142, 154-157


@pytest.mark.parametrize("bits,K,LBp", [(512, 19, 29), (1024, 38, 28), (1536, 56, 28)])
def test_product_scanning_form_bounds_with_sixteen_fold_headroom(bits, K, LBp):
    """csrc/hensel_ps.hpp keeps a residue in K limbs of LB bits with R = 2^(LB K) >= 16 P only (capi_keys.inc: build_hensel;
    the multi-lane forms keep 256).  Upper bounds, as multiples of P, pushed through the kernel's flow with the WORST
    headroom R = 16 P: the chunked entry (a chunk below R/4, at most 4 chunks summed), the window-table recurrence, the main
    loop (squarings, products by any table entry), and the column sums of the widest product.  Everything must stay below
    R (the limbs hold it) and a column below 2^64 (one 64-bit accumulator holds it).  Lazy Montgomery product:
    redc(T) < T/R + P."""
    F = lambda a, b=1: a / b * (1 + 1e-12)      # upper bounds in floating point, rounded up a little at every step
    assert LBp * K >= bits + LBp + 4 and LBp * (K - 1) < bits + LBp + 4          # the K build_hensel picks
    rho = 16.0                                                                   # R / P, worst case
    pair_l2 = {512: 2 * 19, 1024: 4 * 18, 1536: 8 * 14}[bits]                                 # 29-bit limbs per half of an n^2 pair row
    chunks = -(-pair_l2 // ((LBp * K - 2) // 29))                                # (build_hensel_set: pchunks)
    assert chunks <= 4

    def mul(c1, c2):                       # one lazy product of values below c1 P and c2 P
        return F(c1 * c2, rho) + 1

    def pairmul(x, y):                     # (a, b) (x) (c, d): t = a c, w = a d + b c + q (q < R: + 1 P at most, counted in the +1)
        (a, b), (c, d) = x, y
        return (mul(a, c), F(a * d + b * c, rho) + 1 + 1e-6)

    # entry: chunk values below R/4 = rho/4 P, conversion constants below P
    z = rho / 4
    a, b = pairmul((z, 0.0), (1.0, 1.0))
    b += mul(z, 1.0)                      # ps_add(b, tb)
    base = (chunks * a, chunks * b)        # ps_add(acc, .) over the chunks
    assert max(base) < rho
    # window table: entry e = entry e-1 (x) base; the bounds are monotone in the inputs, so their running maximum covers all
    entry = base
    worst = base
    for _ in range(64):
        entry = pairmul(entry, base)
        worst = (max(worst[0], entry[0]), max(worst[1], entry[1]))
    assert max(worst) < rho
    # main loop from any entry: squarings and products by the worst entry, to a fixed point
    s = worst
    top = s
    for _ in range(200):
        for _ in range(5):
            s = pairmul(s, s)
            top = (max(top[0], s[0]), max(top[1], s[1]))
        s = pairmul(s, worst)
        top = (max(top[0], s[0]), max(top[1], s[1]))
    assert max(top) < rho and max(s) < 3
    # columns: canonical limbs below 2^LB, the doubled operand of a squaring below 2^(LB+1): at most 3K products of 2^(2 LB) each
    assert 3 * K * (1 << (2 * LBp)) < 1 << 64



@pytest.mark.parametrize("pbits,K,LBk,wide", [(1024, 38, 28, False), (1024, 38, 28, True), (512, 19, 29, True), (1536, 56, 28, False)])
def test_wavefront_wide_form_lane_model(pbits, K, LBk, wide):
    """csrc/hensel_wave.hpp lane for lane with Python integers: one limb per lane, operand scan with the accumulators sliding
    down the lanes, the two scans of a pair product in lock-step, relaxed limbs, and -- `wide` -- whole-word (32-bit) quotient
    digits where R >= 2^10 P.  Squarings and products from worst-case inputs must give the pair product of hensel.hpp, every
    accumulator must stay below 2^64, and the values must settle below the bound the exit relies on (2 P, or 17 P / 9 P with
    wide digits) and fit the K limbs."""
    M = (1 << LBk) - 1
    rng = random.Random(pbits + K + wide)
    while True:                                     # a prime-like odd modulus is enough for the algebra: any odd p works
        p = rng.getrandbits(pbits) | (1 << (pbits - 1)) | 1
        if p % 3 and p % 5:
            break
    k = (-pow(p, -1, 1 << LBk)) % (1 << LBk)
    P = p * k
    R = 1 << (LBk * K)
    assert P % (1 << LBk) == M and R >= 16 * P and (not wide or LBk * K - (pbits + LBk) >= 10)      # capi.cpp: decrypt_on
    nl = [(P >> (LBk * l)) & M for l in range(K)] + [0]
    worst = [0]

    def lanes(v):                                   # canonical limbs, lane K is the zero lane above them
        return [(v >> (LBk * l)) & M for l in range(K)] + [0]

    def num(ls):
        return sum(v << (LBk * l) for l, v in enumerate(ls))

    def slide(acc):
        return [(acc[l + 1] & M if l + 1 <= K else 0) + (acc[l] >> LBk) for l in range(K + 1)]

    def digit(acc):
        return acc[0] & (0xFFFFFFFF if wide else M)

    def finish(acc):
        return [(acc[l] & M) + ((acc[l - 1] >> LBk) if l else 0) for l in range(K + 1)]

    def pairop(a, b, c, d, sqr):
        """wv_pairsqr (c = a, d = b, the b operand doubled) / wv_pairmul, lock-step"""
        acc1, acc2 = [0] * (K + 1), [0] * (K + 1)
        mul2 = [2 * v for v in b] if sqr else d
        for i in range(K):
            acc1 = [acc1[l] + a[i] * c[l] for l in range(K + 1)]
            q1 = digit(acc1)
            acc2 = [acc2[l] + a[i] * mul2[l] for l in range(K + 1)]
            if not sqr:
                acc2 = [acc2[l] + b[i] * c[l] for l in range(K + 1)]
            acc1 = [acc1[l] + q1 * nl[l] for l in range(K + 1)]
            acc2[0] += q1
            q2 = digit(acc2)
            worst[0] = max(worst[0], max(acc1), max(acc2) + q2 * M)
            assert acc1[0] & M == 0
            acc1 = slide(acc1)
            acc2 = [acc2[l] + q2 * nl[l] for l in range(K + 1)]
            assert acc2[0] & M == 0
            acc2 = slide(acc2)
        t, w = finish(acc1), finish(acc2)
        assert t[K] == 0 and w[K] == 0
        return t, w

    bound = ((1 << (32 - LBk)) + 1) * P if wide else 2 * P
    start = 4 * P - 1 if not wide else bound - 1              # the largest values the flow can hand in
    a, b = lanes(start), lanes(start - 12345)
    c, d = lanes(start - 99), lanes(start - 7)
    x, y = (num(a), num(b)), (num(c), num(d))
    for rounds in range(3):
        t, w = pairop(a, b, c, d, False)
        assert (num(t) - P * num(w)) % (P * P) == (x[0] - P * x[1]) * (y[0] - P * y[1]) * pow(R, -1, P * P) % (P * P)
        assert num(t) < bound and num(w) < bound and max(t + w) < (1 << LBk) + (1 << 9)
        a, b = t, w
        x = (num(a), num(b))
        t, w = pairop(a, b, a, b, True)
        assert (num(t) - P * num(w)) % (P * P) == (x[0] - P * x[1]) ** 2 * pow(R, -1, P * P) % (P * P)
        assert num(t) < bound and num(w) < bound
        a, b = t, w
        x = (num(a), num(b))
    assert worst[0] < 1 << 64
