"""CPU oracle for the Paillier hot path -- TEST INFRASTRUCTURE ONLY.

A restatement, in arbitrary-precision Python integers, of what intel/pailliercryptolib
(IPCL v2.0.0) computes on its batched-modexp hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and only as the checker.  The product path (``pailliercryptolib_amd``) never imports it.

Pinning: IPCL's arithmetic lives in IPP-Crypto (tag ippcp_2021.6, not vendored, cannot be
built here), so this oracle is pinned against the reference's own known-answer test
``CryptoTest.ISO_IEC_18033_6_ComplianceTest`` (test/test_cryptography.cpp:99-241): see
``tests/golden/iso_kat.json`` and ``tests/test_oracle.py``.  DJN encrypt, CT*PT, RAW decrypt
and key sizes other than 2048 are not pinned by any reference vector ("parity unpinned" for
those rows); they rest on modexp being a mathematical function, cross-checked between this
oracle (CPython ``pow``), the independent C restatement in ``oracle/modexp_oracle.c`` and,
where libcrypto exists, OpenSSL ``BN_mod_exp`` (the oracle the reference's own QAT tests
use, module/heqat/test/test_bnModExp.cpp:57-60).

All citations are file:line under the reference tree.
"""
from dataclasses import dataclass
from math import gcd
from typing import List, Optional, Sequence


def mod_exp(base: int, exp: int, mod: int) -> int:
    """ipcl::modExp(BigNumber, BigNumber, BigNumber) -- ipcl/mod_exp.cpp:739-749.

    The reference does not pre-reduce the base (callers guarantee base < mod,
    SURVEY Appendix A Q10); mathematically the result is the same, so the oracle reduces.
    """
    return pow(base % mod, exp, mod)


def mod_exp_batch(base: Sequence[int], exp: Sequence[int], mod: Sequence[int]) -> List[int]:
    """ipcl::modExp(vector, vector, vector) -- ipcl/mod_exp.cpp:680-737; element-wise,
    sizes must agree (ippMBModExp ERROR_CHECK, mod_exp.cpp:452-454)."""
    if not (len(base) == len(exp) == len(mod)):
        raise RuntimeError("modExp: input vector size error")
    return [mod_exp(b, e, m) for b, e, m in zip(base, exp, mod)]


@dataclass
class PublicKey:
    """ipcl::PublicKey -- ipcl/pub_key.cpp:18-49 (ctor/enableDJN), 131-162 (create)."""
    n: int
    bits: int
    djn: bool = False
    hs: int = 0
    randbits: int = 0

    @property
    def nsq(self) -> int:               # pub_key.cpp:21
        return self.n * self.n

    @property
    def g(self) -> int:                 # pub_key.cpp:20
        return self.n + 1

    def set_djn(self, hs: int, randbits: Optional[int] = None) -> None:
        """setDJN / setHS -- pub_key.cpp:131-137, 97; randbits = bits/2 (pub_key.cpp:46)."""
        self.hs = hs
        self.randbits = self.bits >> 1 if randbits is None else randbits
        self.djn = True

    def hs_from_x(self, x: int) -> int:
        """enableDJN's hs for a given random x -- pub_key.cpp:41-45:
        h = (-x^2) mod n (non-negative residue), hs = h^n mod n^2."""
        if gcd(x, self.n) != 1:
            raise ValueError("x must be coprime to n (pub_key.cpp:34-39)")
        h = (-(x % self.n) ** 2) % self.n
        return pow(h, self.n, self.nsq)

    def raw_encrypt_no_obf(self, m: int) -> int:
        """(n*m + 1) % n^2 -- pub_key.cpp:105 (g^m with g = n+1)."""
        return (self.n * m + 1) % self.nsq

    def obfuscator(self, r: int) -> int:
        """getDJNObfuscator hs^r mod n^2 (pub_key.cpp:51-64) or getNormalObfuscator
        r^n mod n^2 (pub_key.cpp:66-80).  r is used as injected: not reduced, not truncated
        (setRandom, pub_key.cpp:56-57,71-72,92-95)."""
        if self.djn:
            return pow(self.hs, r, self.nsq)
        return pow(r, self.n, self.nsq)

    def encrypt(self, m: Sequence[int], r: Optional[Sequence[int]] = None,
                make_secure: bool = True) -> List[int]:
        """PublicKey::encrypt / raw_encrypt / applyObfuscator -- pub_key.cpp:82-129."""
        if len(m) == 0:
            raise RuntimeError("encrypt: Cannot encrypt empty PlainText")   # pub_key.cpp:116
        ct = [self.raw_encrypt_no_obf(x) for x in m]
        if make_secure:
            if r is None or len(r) != len(m):
                raise RuntimeError("modExp: input vector size error")        # mod_exp.cpp:452-454
            ct = [(c * self.obfuscator(ri)) % self.nsq for c, ri in zip(ct, r)]  # pub_key.cpp:88-89
        return ct


@dataclass
class PrivateKey:
    """ipcl::PrivateKey -- ipcl/pri_key.cpp:13-63 (precompute), 65-167."""
    n: int
    p: int
    q: int

    def __post_init__(self):
        if self.q < self.p:                       # pri_key.cpp:19-22: p < q
            self.p, self.q = self.q, self.p
        if self.p * self.q != self.n:
            raise RuntimeError("PrivateKey ctor: Public key does not match p * q.")
        if self.p == self.q:
            raise RuntimeError("PrivateKey ctor: p and q are same")
        self.nsq = self.n * self.n
        self.g = self.n + 1
        self.pm1, self.qm1 = self.p - 1, self.q - 1        # pri_key.cpp:23-24
        self.psq, self.qsq = self.p * self.p, self.q * self.q
        self.pinv = pow(self.p, -1, self.q)                # pri_key.cpp:27  q.InverseMul(p)
        self.hp = self._hfun(self.p, self.psq)             # pri_key.cpp:28
        self.hq = self._hfun(self.q, self.qsq)
        self.lam = self.pm1 * self.qm1 // gcd(self.pm1, self.qm1)   # pri_key.hpp:23-27
        self.x = pow((pow(self.g, self.lam, self.nsq) - 1) // self.n, -1, self.n)  # pri_key.cpp:31-32

    @staticmethod
    def _lfun(a: int, b: int) -> int:
        return (a - 1) // b                                # pri_key.cpp:154-157

    def _hfun(self, a: int, b: int) -> int:
        """computeHfun -- pri_key.cpp:159-167."""
        pm = pow(self.g % b, a - 1, b)
        return pow(self._lfun(pm, a), -1, a)

    def decrypt_crt(self, ct: Sequence[int]) -> List[int]:
        """decryptCRT + computeCRT -- pri_key.cpp:114-152."""
        out = []
        for c in ct:
            resp = pow(c % self.psq, self.pm1, self.psq)          # pri_key.cpp:128,133
            resq = pow(c % self.qsq, self.qm1, self.qsq)          # pri_key.cpp:129,134
            dp = self._lfun(resp, self.p) * self.hp % self.p      # pri_key.cpp:142
            dq = self._lfun(resq, self.q) * self.hq % self.q      # pri_key.cpp:143
            u = (dq - dp) * self.pinv % self.q                    # pri_key.cpp:150 (non-negative residue)
            out.append(dp + u * self.p)                           # pri_key.cpp:151
        return out

    def decrypt_raw(self, ct: Sequence[int]) -> List[int]:
        """decryptRAW -- pri_key.cpp:92-111."""
        return [((pow(c, self.lam, self.nsq) - 1) // self.n) * self.x % self.n for c in ct]

    def decrypt(self, ct: Sequence[int], crt: bool = True) -> List[int]:
        if len(ct) == 0:
            raise RuntimeError("decrypt: Cannot decrypt empty CipherText")   # pri_key.cpp:71
        return self.decrypt_crt(ct) if crt else self.decrypt_raw(ct)


def ct_add(a: Sequence[int], b: Sequence[int], nsq: int) -> List[int]:
    """CipherText::operator+(CT) / raw_add -- ciphertext.cpp:35-72,135-141.
    b may have size 1 (scalar broadcast)."""
    if not (len(a) == len(b) or len(b) == 1):
        raise RuntimeError("CT + CT error: Size mismatch!")
    return [x * (b[0] if len(b) == 1 else b[i]) % nsq for i, x in enumerate(a)]


def ct_mul_pt(a: Sequence[int], pt: Sequence[int], nsq: int) -> List[int]:
    """CipherText::operator*(PT) / raw_mul -- ciphertext.cpp:83-106,143-162: a^pt mod n^2."""
    if not (len(a) == len(pt) or len(pt) == 1):
        raise RuntimeError("CT * PT error: Size mismatch!")
    return [pow(x, (pt[0] if len(pt) == 1 else pt[i]), nsq) for i, x in enumerate(a)]


# ---- limb helpers shared by tests (little-endian u64 limbs, the C-ABI layout) ----
def to_limbs(x: int, nlimbs: int) -> List[int]:
    if x < 0 or x >> (64 * nlimbs):
        raise ValueError("value does not fit")
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs)]


def from_limbs(limbs: Sequence[int]) -> int:
    v = 0
    for i, w in enumerate(limbs):
        v |= int(w) << (64 * i)
    return v
