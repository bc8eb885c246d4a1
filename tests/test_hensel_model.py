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


def test_one_lane_n2_form_shifted_radix_three_scans():
    """csrc/hensel_ps_n2.hpp, limb for limb with Python integers: the SAME loop modulus P = n k (k = -n^-1 mod 2^29, so P == -1
    modulo 2^28 as well) scanned in K = 75 limbs of 28 bits; the row value x (Montgomery radix R = 2^(29*72)) carried as
    x~ = x 2^s, s = 28*75 - 29*72 = 12; the general pair product as three column scans (t = a c with its digits; U = a d + q
    unreduced; b = b c + U reduced); the exit as two s-bit digit steps.  CT x PT with a fixed window must give pow(c, e, n^2)
    as a pair row of the 29-bit domain; every value stays below 2^(28*75) and every column below 2^64."""
    LBk, K, L2, RB = 28, 75, 72, 29
    M = (1 << LBk) - 1
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q = int(k["p"], 16), int(k["q"], 16)
    n = p * q
    nsq = n * n
    kk = (-pow(n, -1, 1 << RB)) % (1 << RB)
    P = n * kk
    assert P % (1 << RB) == (1 << RB) - 1 and P % (1 << LBk) == M
    R, Rp = 1 << (RB * L2), 1 << (LBk * K)
    S = LBk * K - RB * L2
    assert S == 12 and n.bit_length() + RB + 3 + S <= LBk * K          # capi.cpp: modexp_ps_applies
    limbs = lambda v, cnt: [(v >> (LBk * i)) & M for i in range(cnt)]
    num = lambda ls: sum(v << (LBk * i) for i, v in enumerate(ls))
    nl = limbs(P, K)
    n1p = nl[1] + 1
    colmax = [0]

    def scan(x, y, addend, reduce):
        """one column scan: sum x_i y_(col-i) (+ addend[col]) (+ q_i n_(col-i) with unit digits when reduce); returns
        (result limbs, digits)"""
        acc, qd, out = 0, [], []
        for col in range(2 * K):
            lo, hi = max(0, col - K + 1), min(col, K - 1)
            if reduce:
                for i in range(lo, min(col - 1, K - 1) + 1):          # q_i n_(col-i), i <= col - 1; n_0 rides on n1p
                    acc += qd[i] * (n1p if col - i == 1 else nl[col - i])
            for i in range(lo, hi + 1):
                acc += x[i] * y[col - i]
            if addend is not None and col < len(addend):
                acc += addend[col]
            colmax[0] = max(colmax[0], acc)
            if reduce and col < K:
                qd.append(acc & M)
            else:
                out.append(acc & M)
            acc >>= LBk
        assert acc == 0
        return out, qd

    def pairmul_mem(x, y):                     # ps_pairmul_mem
        (a, b), (c, d) = x, y
        t, qd = scan(a, c, None, True)         # psn_mul_digits
        U, _ = scan(a, d, qd, False)           # psn_mul_plain: 2K limbs
        assert len(U) == 2 * K
        w, _ = scan(b, c, U, True)             # psn_mul_add
        return (t, w)

    def pairsqr(x):                            # ps_pairsqr: t = a a with digits; b = 2 a b + q reduced
        a, b = x
        t, qd = scan(a, a, None, True)
        w, _ = scan(a, [2 * v for v in b], qd, True)
        return (t, w)

    def pval(pr):                              # the residue a pair of limb vectors stands for
        return (num(pr[0]) - P * num(pr[1])) % (P * P)

    def digit_step(x):                         # psn_digit_step
        v = num(x)
        qq = v & ((1 << S) - 1)
        v2 = v + qq * P
        assert v2 % (1 << S) == 0
        return limbs(v2 >> S, K), qq

    rng = random.Random(75)
    for c, e, w in ((rng.randrange(nsq), rng.getrandbits(33), 3), (nsq - 1, (1 << 17) - 1, 3), (rng.randrange(nsq), 0, 1),
                    (0, 5, 2), (rng.randrange(nsq), rng.getrandbits(64), 4)):
        # A row as a multi-lane kernel may leave it: components up to 4P.  (+3P, +3P) is NOT another representative of c R --
        # it is the pair of another residue, c_eff R -- but it has the largest components a row can have, which is what the
        # bounds are about; the kernel's contract is on the VALUE a - P b of the row, so that value is the base here.
        a0, b0 = to_pair(c * R, P)
        a0, b0 = a0 + 3 * P, b0 + 3 * P
        c_eff = (a0 - P * b0) * pow(R, -1, nsq) % nsq
        base = ((a0 << S, b0 << S))                            # psn_relimb_in with the shift
        assert max(base) < 1 << (LBk * K)
        x1 = (limbs(base[0], K), limbs(base[1], K))
        one = to_pair(R, P)
        tbl = [(limbs(one[0] << S, K), limbs(one[1] << S, K)), x1]
        for _ in range(2, 1 << w):
            tbl.append(pairmul_mem(tbl[-1], x1))
        nwin = (max(e.bit_length(), 1) + w - 1) // w
        x = tbl[(e >> (w * (nwin - 1))) & ((1 << w) - 1)]
        for i in range(nwin - 2, -1, -1):
            for _ in range(w):
                x = pairsqr(x)
            x = pairmul_mem(x, tbl[(e >> (w * i)) & ((1 << w) - 1)])
        assert pval(x) % nsq == pow(c_eff, e, nsq) * R * (1 << S) % nsq          # x~ = (c^e R) 2^s
        # exit: ps_pair_shift_out
        a1, q1 = digit_step(x[0])
        b1 = limbs(num(x[1]) + q1, K)
        b2, _ = digit_step(b1)
        out = (num(a1), num(b2))
        assert max(out) < 1 << (RB * L2)                                          # fits the 72 limbs of 29 bits of a row
        assert (out[0] - P * out[1]) % nsq == pow(c_eff, e, nsq) * R % nsq         # the row of c^e
    assert colmax[0] < 1 << 64
