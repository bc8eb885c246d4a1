"""Device-resident batch operations over torch tensors (plumbing for bench / large tests).

A batch is a contiguous int64 CUDA tensor [count, words] holding little-endian 64-bit limbs (the
C-ABI layout, include/pgpu.h); every call only enqueues kernels on torch's current stream.
"""
import ctypes

import numpy as np
import torch

from . import _capi
from .limbs import ints_to_limbs


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_batch(t, name):
    if not (t.is_cuda and t.dtype == torch.int64 and t.dim() == 2 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous CUDA int64 tensor [count, words]")


def to_device(limbs_u64):
    return torch.from_numpy(np.ascontiguousarray(limbs_u64, dtype=np.uint64).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def encrypt(pk, d_m, d_r, r_bits=None):
    """PublicKey.encrypt on device buffers: d_m [n, mw], d_r [n, rw] -> [n, 2*n_words]."""
    _check_batch(d_m, "encrypt m"); _check_batch(d_r, "encrypt r")
    if d_m.shape[0] == 0:
        raise RuntimeError("encrypt: Cannot encrypt empty PlainText")
    if d_m.shape[0] != d_r.shape[0]:
        raise RuntimeError("modExp: input vector size error")
    out = torch.empty((d_m.shape[0], 2 * pk.n_words), dtype=torch.int64, device=d_m.device)
    rb = 64 * d_r.shape[1] if r_bits is None else int(r_bits)
    _capi.check(_capi.lib().pgpu_paillier_encrypt_dev(pk._h, d_m.data_ptr(), d_m.shape[1], d_m.shape[1],
                                                      d_r.data_ptr(), d_r.shape[1], d_r.shape[1], rb,
                                                      out.data_ptr(), d_m.shape[0], _stream()))
    return out


def decrypt(sk, d_c):
    _check_batch(d_c, "decrypt c")
    if d_c.shape[0] == 0:
        raise RuntimeError("decrypt: Cannot decrypt empty CipherText")
    if d_c.shape[1] != 2 * sk.n_words:
        raise RuntimeError("decrypt: ciphertext width mismatch")
    out = torch.empty((d_c.shape[0], sk.n_words), dtype=torch.int64, device=d_c.device)
    _capi.check(_capi.lib().pgpu_paillier_decrypt_crt_dev(sk._h, d_c.data_ptr(), out.data_ptr(), d_c.shape[0],
                                                          _stream()))
    return out


def mod_mul(d_a, d_b, mod):
    """a[i]*b[i] mod `mod` (Python int); d_b may have one row (scalar broadcast)."""
    _check_batch(d_a, "modmul a"); _check_batch(d_b, "modmul b")
    W = d_a.shape[1]
    if d_b.shape[1] != W or d_b.shape[0] not in (1, d_a.shape[0]):
        raise RuntimeError("CT + CT error: Size mismatch!")
    h_mod = ints_to_limbs([mod], W)[0]
    out = torch.empty_like(d_a)
    _capi.check(_capi.lib().pgpu_modmul_dev(d_a.data_ptr(), d_b.data_ptr(), W if d_b.shape[0] == d_a.shape[0] else 0,
                                            h_mod.ctypes.data_as(ctypes.c_void_p), W, out.data_ptr(),
                                            d_a.shape[0], _stream()))
    return out


def mod_exp(d_base, d_exp, mod, exp_bits=None):
    """base[i]^exp[i] mod `mod`; d_base / d_exp may have one row (shared)."""
    _check_batch(d_base, "modexp base"); _check_batch(d_exp, "modexp exp")
    W, E = d_base.shape[1], d_exp.shape[1]
    n = max(d_base.shape[0], d_exp.shape[0])
    if d_base.shape[0] not in (1, n) or d_exp.shape[0] not in (1, n):
        raise RuntimeError("modExp: input vector size error")
    h_mod = ints_to_limbs([mod], W)[0]
    out = torch.empty((n, W), dtype=torch.int64, device=d_base.device)
    _capi.check(_capi.lib().pgpu_modexp_dev(d_base.data_ptr(), W if d_base.shape[0] == n else 0, d_exp.data_ptr(),
                                            E if d_exp.shape[0] == n else 0, E, 64 * E if exp_bits is None else exp_bits,
                                            h_mod.ctypes.data_as(ctypes.c_void_p), W, out.data_ptr(), n, _stream()))
    return out
