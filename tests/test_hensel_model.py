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

