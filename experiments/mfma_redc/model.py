"""Byte-radix Montgomery reduction as two GEMMs against matrices the whole batch shares -- CPU model.

Groundwork for DESIGN.md section 8 item 5 (not part of the product).  For an odd modulus N and R = 256^D:
    Q = (T mod R) * N' mod R,   N' = -N^-1 mod R          -> GEMM 1: lower-triangular Toeplitz(N') x bytes(T_lo)
    U = (T + Q*N) / R                                      -> GEMM 2: Toeplitz(N) x bytes(Q), columns >= D-5 only
The MFMA multiplies signed bytes, so the shared matrices are written in balanced digits (-128..127) and the
data bytes are offset by 128, with the correction 128 * rowsum folded into the accumulator's start value.
The carry out of the low half needs no low columns: the low half of T + Q*N is a multiple of R, so it
follows from the top five columns (integer rounding of a fixed-point sum).
"""
import numpy as np


def balanced_digits(v, count):
    """v = sum d_i 256^i with d_i in [-128, 127]; `count` digits (v must fit)."""
    out = []
    for _ in range(count):
        d = v & 0xFF
        if d >= 128:
            d -= 256
        out.append(d)
        v = (v - d) >> 8
    assert v == 0, "value does not fit the balanced digits"
    return np.array(out, dtype=np.int64)


def to_bytes(v, count):
    return np.array([(v >> (8 * i)) & 0xFF for i in range(count)], dtype=np.int64)


class RedcModel:
    TOP = 5      # columns of the low half that determine its carry

    def __init__(self, N, D):
        assert N & 1 and N < 256 ** D
        self.N, self.D = N, D
        self.R = 256 ** D
        self.Np = (-pow(N, -1, self.R)) % self.R
        # balanced digits; one extra digit absorbs the final carry of the balanced form
        self.np_d = balanced_digits(self.Np if self.Np < self.R // 2 else self.Np - self.R, D + 1)[:D]
        # (N' is only needed mod R, so its balanced form may represent N' - R)
        self.n_d = balanced_digits(N, D + 1)
        D1 = D
        # GEMM 1: A1[i][k] = np_d[i-k], 0 <= k <= i < D
        self.A1 = np.zeros((D1, D), dtype=np.int64)
        for i in range(D1):
            for k in range(i + 1):
                self.A1[i, k] = self.np_d[i - k]
        # GEMM 2: rows c = D-TOP .. 2D, A2[c][k] = n_d[c-k], 0 <= c-k <= D
        self.c0 = D - self.TOP
        rows = 2 * D + 1 - self.c0
        self.A2 = np.zeros((rows, D), dtype=np.int64)
        for r in range(rows):
            c = self.c0 + r
            for k in range(D):
                if 0 <= c - k <= D:
                    self.A2[r, k] = self.n_d[c - k]
        self.corr1 = 128 * self.A1.sum(axis=1)
        self.corr2 = 128 * self.A2.sum(axis=1)

    def reduce(self, T):
        """T < N * R.  Returns U = (T + Q*N)/R computed the way the kernel does."""
        D, R = self.D, self.R
        t = to_bytes(T, 2 * D)
        # GEMM 1 on offset data bytes
        s1 = self.A1 @ (t[:D] - 128) + self.corr1            # column sums of T_lo * N'  (exact integers)
        q_val = sum(int(s1[i]) << (8 * i) for i in range(D)) % R
        q = to_bytes(q_val, D)
        assert q_val == (T % R) * self.Np % R
        # GEMM 2
        s2 = self.A2 @ (q - 128) + self.corr2                # column sums c0 .. 2D of Q*N
        # carry of the low half from its top columns: (sum_{c<D} (P_c + t_c) 256^c) / R is an integer m
        V = sum((int(s2[self.TOP - j]) + int(t[D - j])) << (8 * (self.TOP - j)) for j in range(1, self.TOP + 1))
        m = (V + (1 << (8 * self.TOP - 1))) >> (8 * self.TOP)
        U = m + sum((int(s2[self.TOP + c]) + (int(t[D + c]) if c < D else 0)) << (8 * c) for c in range(D + 1))
        return U


if __name__ == "__main__":
    import random
    rng = random.Random(3)
    for bits, D in ((2048, 264), (2088, 264), (1024, 136), (4096, 520)):
        N = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        mdl = RedcModel(N, D)
        for _ in range(20):
            T = rng.randrange(N * mdl.R) if rng.random() < 0.8 else rng.choice([0, 1, N * mdl.R - 1, mdl.R - 1, mdl.R])
            U = mdl.reduce(T)
            Q = (T % mdl.R) * mdl.Np % mdl.R
            assert U == (T + Q * N) // mdl.R and (T + Q * N) % mdl.R == 0, (bits, D)
            assert U < 2 * N and (U * mdl.R - T) % N == 0
    print("model ok")
