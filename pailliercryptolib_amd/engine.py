"""Host-side Python mirror of the hot-path seam: batched modExp / modMul on the GPU.

Mirrors ipcl::modExp(vector, vector, vector) (reference ipcl/mod_exp.cpp:680-737) and
CipherText::raw_add (ciphertext.cpp:135-141) over numpy / torch buffers.  Every function calls
the C-ABI (include/pgpu.h); nothing here computes on the CPU.
"""
import ctypes

import numpy as np

from . import _capi
from .limbs import ints_to_limbs, limbs_to_ints

_initialized = False


def initialize(device=None):
    """ipcl::initializeContext counterpart (utils/context.cpp:40-55): bind this process to one GPU."""
    global _initialized
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if torch is going to
    # share this process it must bring the runtime up first (the library then binds to the same one).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = _capi.lib()
    _capi.check(L.pgpu_init(-1 if device is None else int(device)))
    _initialized = True


def terminate():
    global _initialized
    if _capi._lib is not None:
        _capi.lib().pgpu_shutdown()
    _initialized = False


def _ensure():
    if not _initialized:
        initialize()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def mod_exp_limbs(base, exp, mod, exp_bits=None):
    """base: [n, W] or [1, W] (shared); exp: [n, E] or [1, E] (shared); mod: [W] -> out [n, W]."""
    _ensure()
    base = np.ascontiguousarray(base, dtype=np.uint64)
    exp = np.ascontiguousarray(exp, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64).reshape(-1)
    W = mod.shape[0]
    if base.ndim != 2 or exp.ndim != 2 or base.shape[1] != W:
        raise RuntimeError("modExp: input vector size error")
    n = max(base.shape[0], exp.shape[0])
    if base.shape[0] not in (1, n) or exp.shape[0] not in (1, n):
        raise RuntimeError("modExp: input vector size error")   # mod_exp.cpp:452-454
    if exp_bits is None:
        exp_bits = max((int(v).bit_length() for v in limbs_to_ints(exp)), default=0)
    out = np.empty((n, W), dtype=np.uint64)
    bs = W if base.shape[0] == n else 0            # stride 0 == one shared value
    es = exp.shape[1] if exp.shape[0] == n else 0
    _capi.check(_capi.lib().pgpu_modexp(_ptr(base), bs, _ptr(exp), es, exp.shape[1], int(exp_bits),
                                        _ptr(mod), W, _ptr(out), n))
    return out


def mod_exp(base, exp, mod):
    """ipcl::modExp over Python ints.  base/exp: lists (len n or 1); mod: one int (shared)."""
    W = (int(mod).bit_length() + 63) // 64
    E = max(1, (max((int(e).bit_length() for e in exp), default=1) + 63) // 64)
    out = mod_exp_limbs(ints_to_limbs([b % (1 << (64 * W)) for b in base], W), ints_to_limbs(exp, E),
                        ints_to_limbs([mod], W)[0])
    return limbs_to_ints(out)


def mod_mul_limbs(a, b, mod):
    """a: [n, W]; b: [n, W] or [1, W] (scalar broadcast); mod: [W]."""
    _ensure()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64).reshape(-1)
    W = mod.shape[0]
    n = a.shape[0]
    if a.shape[1] != W or b.shape[1] != W or b.shape[0] not in (1, n):
        raise RuntimeError("CT + CT error: Size mismatch!")
    out = np.empty((n, W), dtype=np.uint64)
    bstride = W if b.shape[0] == n else 0
    _capi.check(_capi.lib().pgpu_modmul(_ptr(a), _ptr(b), bstride, _ptr(mod), W, _ptr(out), n))
    return out


def mod_mul(a, b, mod):
    W = (int(mod).bit_length() + 63) // 64
    out = mod_mul_limbs(ints_to_limbs(a, W), ints_to_limbs(b, W), ints_to_limbs([mod], W)[0])
    return limbs_to_ints(out)


class PublicKey:
    """Host-side mirror of ipcl::PublicKey's encrypt path (reference ipcl/pub_key.cpp:82-129).

    ``hs`` set -> DJN scheme (obfuscator hs^r); otherwise r^n.  Randomness is supplied by the
    caller (like ``setRandom``, pub_key.cpp:92-95) so results are reproducible.
    """

    def __init__(self, n, bits=None, hs=None):
        _ensure()
        self.n = int(n)
        self.bits = int(bits) if bits is not None else self.n.bit_length()
        self.n_words = (self.n.bit_length() + 63) // 64
        self.hs = None if hs is None else int(hs)
        self.nsq = self.n * self.n
        self._h = ctypes.c_void_p()
        n_l = ints_to_limbs([self.n], self.n_words)
        hs_l = None if hs is None else ints_to_limbs([self.hs], 2 * self.n_words)
        _capi.check(_capi.lib().pgpu_pubkey_create(_ptr(n_l), self.n_words,
                                                   None if hs_l is None else _ptr(hs_l),
                                                   ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _capi._lib is not None:
            _capi._lib.pgpu_pubkey_destroy(self._h)
            self._h = None

    def encrypt_limbs(self, m, r, r_bits=None):
        """m: [n, Wm] uint64, r: [n, Wr] uint64 -> ciphertexts [n, 2*n_words]."""
        m = np.ascontiguousarray(m, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        if m.shape[0] == 0:
            raise RuntimeError("encrypt: Cannot encrypt empty PlainText")     # pub_key.cpp:116
        if r.shape[0] != m.shape[0]:
            raise RuntimeError("modExp: input vector size error")             # mod_exp.cpp:452-454
        if r_bits is None:
            r_bits = max(int(v).bit_length() for v in limbs_to_ints(r))
        out = np.empty((m.shape[0], 2 * self.n_words), dtype=np.uint64)
        _capi.check(_capi.lib().pgpu_paillier_encrypt(self._h, _ptr(m), m.shape[1], m.shape[1], _ptr(r),
                                                      r.shape[1], r.shape[1], int(r_bits), _ptr(out),
                                                      m.shape[0]))
        return out

    def encrypt(self, m, r):
        mw = max(1, (max(int(v).bit_length() for v in m) + 63) // 64) if len(m) else 1
        rw = max(1, (max(int(v).bit_length() for v in r) + 63) // 64) if len(r) else 1
        if len(m) == 0:
            raise RuntimeError("encrypt: Cannot encrypt empty PlainText")
        return limbs_to_ints(self.encrypt_limbs(ints_to_limbs(m, mw), ints_to_limbs(r, rw)))


class PrivateKey:
    """Host-side mirror of ipcl::PrivateKey::decrypt (CRT path, pri_key.cpp:65-90,114-157)."""

    def __init__(self, p, q):
        _ensure()
        self.p, self.q = (int(p), int(q)) if int(p) < int(q) else (int(q), int(p))
        self.n = self.p * self.q
        self.n_words = (self.n.bit_length() + 63) // 64
        pw = (max(self.p.bit_length(), self.q.bit_length()) + 63) // 64
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().pgpu_privkey_create(_ptr(ints_to_limbs([self.p], pw)),
                                                    _ptr(ints_to_limbs([self.q], pw)), pw,
                                                    ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _capi._lib is not None:
            _capi._lib.pgpu_privkey_destroy(self._h)
            self._h = None

    def decrypt_limbs(self, c):
        c = np.ascontiguousarray(c, dtype=np.uint64)
        if c.shape[0] == 0:
            raise RuntimeError("decrypt: Cannot decrypt empty CipherText")     # pri_key.cpp:71
        if c.shape[1] != 2 * self.n_words:
            raise RuntimeError("decrypt: ciphertext width mismatch")
        out = np.empty((c.shape[0], self.n_words), dtype=np.uint64)
        _capi.check(_capi.lib().pgpu_paillier_decrypt_crt(self._h, _ptr(c), _ptr(out), c.shape[0]))
        return out

    def decrypt(self, c):
        if len(c) == 0:
            raise RuntimeError("decrypt: Cannot decrypt empty CipherText")
        return limbs_to_ints(self.decrypt_limbs(ints_to_limbs(c, 2 * self.n_words)))
