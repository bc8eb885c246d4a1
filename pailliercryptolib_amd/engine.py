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
