"""CPU suite: the C-ABI shared library loads and exports every symbol include/pgpu.h declares;
without a GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pgpu_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from pailliercryptolib_amd import _capi, build
    build.build_pgpu()
    L = ctypes.CDLL(_capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_capi.SYMBOLS) == syms       # the Python binding covers the whole header


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pailliercryptolib_amd as pa
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    assert L.pgpu_device_count() == 0
    with pytest.raises(_capi.PgpuError):
        pa.initialize()
    a = np.ones((1, 1), dtype=np.uint64)
    out = np.zeros((1, 1), dtype=np.uint64)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    rc = L.pgpu_modexp(p(a), 1, p(a), 1, 1, 1, p(a), 1, p(out), 1)
    assert rc == -4 and b"pgpu_init" in L.pgpu_last_error()      # PGPU_ERR_NO_DEVICE
    assert out[0, 0] == 0


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pailliercryptolib_amd/ or include/ refers to it."""
    for base in ("pailliercryptolib_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    src = open(os.path.join(d, f), errors="replace").read()
                    assert "oracle" not in src.lower() or f == "build.py", (d, f)


def test_kernel_geometry_query_is_host_only():
    """pgpu_kernel_geometry needs no device: the launch rules (latency / base / wide lane split) as documented
    in include/pgpu.h and DESIGN.md section 3."""
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    g, k = ctypes.c_int(), ctypes.c_int()

    def geo(words, bits, count):
        assert L.pgpu_kernel_geometry(words, bits, count, ctypes.byref(g), ctypes.byref(k)) == 0
        return g.value, k.value
    assert geo(32, 2048, 16) == (16, 5)            # small batch of the 2048-bit class: 16 lanes per element
    assert geo(32, 2048, 4096) == (16, 5)          # ... while it fits one wavefront per SIMD
    assert geo(32, 2048, 6000) == (8, 9)           # base geometry
    assert geo(32, 2048, 16384) == (4, 18)         # wide split from 1024 wavefronts up (the bench's decrypt launch)
    assert geo(64, 4096, 8192) == (8, 18)
    assert geo(64, 4096, 100) == (16, 9)
    assert geo(48, 3072, 64) == (16, 7)
    assert geo(16, 1024, 16384) in ((4, 9), (4, 10))   # its wide split (2,18) would leave half the SIMDs empty; (4,10) = unit quotient digits
    assert L.pgpu_kernel_geometry(200, 12800, 8, ctypes.byref(g), ctypes.byref(k)) != 0   # wider than any geometry


def test_shard_plan_is_a_contiguous_ordered_balanced_cut():
    """Sharding rule of the device pool (host-only query): every element exactly once, in order, shard sizes
    within one of each other, tiny batches on few devices."""
    import ctypes
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    _capi.check(L.pgpu_set_min_shard(256))
    for count in (0, 1, 255, 256, 511, 512, 2100, 8192, 65536, 1000003):
        for pool in (1, 2, 3, 8):
            n = ctypes.c_int()
            b = (ctypes.c_size_t * (pool + 1))()
            _capi.check(L.pgpu_shard_plan(count, pool, ctypes.byref(n), b))
            D = n.value
            assert 1 <= D <= pool and D == max(1, min(pool, count // 256))
            assert b[0] == 0 and b[D] == count
            sizes = [b[i + 1] - b[i] for i in range(D)]
            assert all(s >= 0 for s in sizes) and max(sizes) - min(sizes) <= 1
            assert sorted(sizes, reverse=True) == sizes          # the longer shards come first
    assert L.pgpu_shard_plan(10, 0, None, None) != 0
