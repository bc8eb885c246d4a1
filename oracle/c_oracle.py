"""ctypes loader of the C oracle (oracle/modexp_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def _ensure_built():
    """Fresh checkout: the .so files are build products (git-ignored); build them on first use."""
    if not os.path.exists(os.path.join(_HERE, "libmodexp_oracle.so")):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        _ensure_built()
        name = "libmodexp_oracle_v3.so" if (_cpu_has("bmi2") and _cpu_has("avx2")) else "libmodexp_oracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            path = os.path.join(_HERE, "libmodexp_oracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not built: run `make -C oracle`")
        L = ctypes.CDLL(path)
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.orc_modexp_batch.argtypes = [vp, sz, vp, sz, i, vp, i, vp, sz]
        L.orc_modmul_batch.argtypes = [vp, vp, sz, vp, i, vp, sz]
        L.orc_paillier_encrypt.argtypes = [vp, i, vp, vp, i, vp, i, vp, sz]
        L.orc_paillier_decrypt_crt.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, sz]
        L.orc_paillier_encrypt_finish.argtypes = [vp, i, vp, i, vp, vp, sz]
        L.orc_paillier_decrypt_prepare.argtypes = [vp, vp, i, vp, vp, vp, sz]
        L.orc_paillier_decrypt_finish.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, sz]
        L.orc_max_threads.restype = i
        L.orc_set_threads.argtypes = [i]
        _lib = L
        _lib._path = path
    return _lib


def usable_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def modexp_batch(base, exp, mod):
    """base [n,W], exp [n,E], mod [W] (uint64) -> [n,W]"""
    base = np.ascontiguousarray(base, dtype=np.uint64)
    exp = np.ascontiguousarray(exp, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64)
    out = np.empty_like(base)
    rc = lib().orc_modexp_batch(_p(base), base.shape[1], _p(exp), exp.shape[1], exp.shape[1], _p(mod),
                                mod.shape[0], _p(out), base.shape[0])
    assert rc == 0
    return out


def modmul_batch(a, b, mod):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64)
    out = np.empty_like(a)
    rc = lib().orc_modmul_batch(_p(a), _p(b), b.shape[1] if b.shape[0] == a.shape[0] else 0, _p(mod),
                                mod.shape[0], _p(out), a.shape[0])
    assert rc == 0
    return out


def paillier_encrypt(n, hs, m, r):
    """n [nw], hs [2nw] or None, m [cnt, mw], r [cnt, rw] -> c [cnt, 2nw]"""
    n = np.ascontiguousarray(n, dtype=np.uint64)
    m = np.ascontiguousarray(m, dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    out = np.empty((m.shape[0], 2 * n.shape[0]), dtype=np.uint64)
    hsp = None if hs is None else _p(np.ascontiguousarray(hs, dtype=np.uint64))
    rc = lib().orc_paillier_encrypt(_p(n), n.shape[0], hsp, _p(m), m.shape[1], _p(r), r.shape[1], _p(out),
                                    m.shape[0])
    assert rc == 0
    return out


def paillier_decrypt_crt(p, q, hp, hq, pinv, c):
    """p<q [pw]; hp,hq,pinv [pw]; c [cnt, 4pw] -> m [cnt, 2pw]"""
    arrs = [np.ascontiguousarray(v, dtype=np.uint64) for v in (p, q, hp, hq, pinv)]
    c = np.ascontiguousarray(c, dtype=np.uint64)
    pw = arrs[0].shape[0]
    out = np.empty((c.shape[0], 2 * pw), dtype=np.uint64)
    rc = lib().orc_paillier_decrypt_crt(_p(arrs[0]), _p(arrs[1]), pw, _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
                                        _p(c), _p(out), c.shape[0])
    assert rc == 0
    return out


_ossl = None


def openssl_lib():
    """libopenssl_oracle.so if it was built (libcrypto present at build time), else None."""
    global _ossl
    if _ossl is None:
        _ensure_built()
        path = os.path.join(_HERE, "libopenssl_oracle.so")
        if not os.path.exists(path):
            return None
        try:
            L = ctypes.CDLL(path)
        except OSError:
            return None
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.orc_openssl_modexp_batch.argtypes = [vp, sz, vp, sz, i, vp, i, vp, sz]
        _ossl = L
    return _ossl


def openssl_modexp_batch(base, exp, mod):
    """Same contract as modexp_batch, computed by OpenSSL BN_mod_exp_mont; None when unavailable."""
    L = openssl_lib()
    if L is None:
        return None
    base = np.ascontiguousarray(base, dtype=np.uint64)
    exp = np.ascontiguousarray(exp, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64)
    out = np.empty((base.shape[0], mod.shape[0]), dtype=np.uint64)   # base rows may be wider than the modulus
    rc = L.orc_openssl_modexp_batch(_p(base), base.shape[1], _p(exp), exp.shape[1], exp.shape[1], _p(mod),
                                    mod.shape[0], _p(out), base.shape[0])
    assert rc == 0
    return out


_ifma = None


def ifma_lib():
    """libifma_oracle.so (oracle/ifma_oracle.c) when it was built AND this CPU reports avx512ifma."""
    global _ifma
    if _ifma is None:
        _ensure_built()
        path = os.path.join(_HERE, "libifma_oracle.so")
        if not (os.path.exists(path) and _cpu_has("avx512ifma") and _cpu_has("avx512f")):
            return None
        try:
            L = ctypes.CDLL(path)
        except OSError:
            return None
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.orc_ifma_modexp_batch.argtypes = [vp, sz, vp, sz, i, vp, i, vp, sz]
        if hasattr(L, "orc_ifma_fb_build"):
            L.orc_ifma_fb_build.argtypes = [vp, vp, i, i, i]
            L.orc_ifma_fb_build.restype = vp
            L.orc_ifma_fb_free.argtypes = [vp]
            L.orc_ifma_fb_free.restype = None
            L.orc_ifma_fb_modexp_batch.argtypes = [vp, vp, sz, i, vp, sz]
            L.orc_ifma_fb_modexp_batch.restype = i
        _ifma = L
    return _ifma


def ifma_modexp_batch(base, exp, mod):
    """Same contract as modexp_batch (base < mod required), 8 lanes per zmm; None when unavailable."""
    L = ifma_lib()
    if L is None:
        return None
    base = np.ascontiguousarray(base, dtype=np.uint64)
    exp = np.ascontiguousarray(exp, dtype=np.uint64)
    mod = np.ascontiguousarray(mod, dtype=np.uint64)
    assert base.shape[1] == mod.shape[0]
    out = np.empty_like(base)
    rc = L.orc_ifma_modexp_batch(_p(base), base.shape[1], _p(exp), exp.shape[1], exp.shape[1], _p(mod),
                                 mod.shape[0], _p(out), base.shape[0])
    assert rc == 0, rc
    return out


class IfmaFixedBase:
    """base^exp[i] for ONE base through a fixed-base table (oracle/ifma_oracle.c: orc_ifma_fb_*): the CPU counterpart of the
    GPU's DJN obfuscator; the table is built once (outside any timed region, like the GPU's)."""

    def __init__(self, base, mod, exp_bits, w=8):
        self.L = ifma_lib()
        if self.L is None or not hasattr(self.L, "orc_ifma_fb_build"):
            raise RuntimeError("libifma_oracle.so (with the fixed-base leg) is not available on this host")
        self.mod = np.ascontiguousarray(mod, dtype=np.uint64)
        base = np.ascontiguousarray(base, dtype=np.uint64)
        assert base.shape[0] == self.mod.shape[0]
        self.h = self.L.orc_ifma_fb_build(_p(base), _p(self.mod), self.mod.shape[0], int(exp_bits), int(w))
        if not self.h:
            raise RuntimeError("orc_ifma_fb_build failed")

    def __call__(self, base_ignored, exp, mod_ignored=None):
        exp = np.ascontiguousarray(exp, dtype=np.uint64)
        out = np.empty((exp.shape[0], self.mod.shape[0]), dtype=np.uint64)
        rc = self.L.orc_ifma_fb_modexp_batch(self.h, _p(exp), exp.shape[1], exp.shape[1], _p(out), exp.shape[0])
        if rc != 0:
            raise RuntimeError(f"orc_ifma_fb_modexp_batch failed: {rc}")
        return out

    def close(self):
        if self.h:
            self.L.orc_ifma_fb_free(self.h)
            self.h = None


# ---- the same encrypt / CRT-decrypt flows with the modexps done by another backend -------------------
# backend(base [cnt, W], exp [cnt, E], mod [W]) -> [cnt, W]: ifma_modexp_batch or openssl_modexp_batch;
# the reference's host glue around them (pub_key.cpp:88-89,105; pri_key.cpp:128-157) stays in
# oracle/modexp_oracle.c (orc_paillier_encrypt_finish / _decrypt_prepare / _decrypt_finish).

def paillier_encrypt_with(backend, n, hs, m, r):
    n = np.ascontiguousarray(n, dtype=np.uint64)
    m = np.ascontiguousarray(m, dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    nw, cnt = n.shape[0], m.shape[0]
    nsq = np.zeros(2 * nw, dtype=np.uint64)
    v = _to_int(n) ** 2
    for k in range(2 * nw):
        nsq[k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    if hs is not None:                                   # DJN: hs^r   pub_key.cpp:63
        obf = backend(np.tile(np.ascontiguousarray(hs, dtype=np.uint64), (cnt, 1)), r, nsq)
    else:                                                # r^n        pub_key.cpp:79
        base = np.zeros((cnt, 2 * nw), dtype=np.uint64)
        base[:, :r.shape[1]] = r
        obf = backend(base, np.tile(n, (cnt, 1)), nsq)
    out = np.empty((cnt, 2 * nw), dtype=np.uint64)
    rc = lib().orc_paillier_encrypt_finish(_p(n), nw, _p(m), m.shape[1], _p(obf), _p(out), cnt)
    assert rc == 0
    return out


def paillier_decrypt_crt_with(backend, p, q, hp, hq, pinv, c):
    arrs = [np.ascontiguousarray(v, dtype=np.uint64) for v in (p, q, hp, hq, pinv)]
    c = np.ascontiguousarray(c, dtype=np.uint64)
    pw, cnt = arrs[0].shape[0], c.shape[0]
    bp = np.empty((cnt, 2 * pw), dtype=np.uint64)
    bq = np.empty_like(bp)
    rc = lib().orc_paillier_decrypt_prepare(_p(arrs[0]), _p(arrs[1]), pw, _p(c), _p(bp), _p(bq), cnt)
    assert rc == 0
    res = []
    for prime, b in ((arrs[0], bp), (arrs[1], bq)):
        pv = _to_int(prime)
        e = np.array([((pv - 1) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(pw)], dtype=np.uint64)
        sq = np.array([((pv * pv) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(2 * pw)], dtype=np.uint64)
        res.append(np.ascontiguousarray(backend(b, np.tile(e, (cnt, 1)), sq)))   # pri_key.cpp:133-134
    out = np.empty((cnt, 2 * pw), dtype=np.uint64)
    rc = lib().orc_paillier_decrypt_finish(_p(arrs[0]), _p(arrs[1]), pw, _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
                                           _p(res[0]), _p(res[1]), _p(out), cnt)
    assert rc == 0
    return out


def _to_int(limbs):
    return sum(int(w) << (64 * k) for k, w in enumerate(limbs))
