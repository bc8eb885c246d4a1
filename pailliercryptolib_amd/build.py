"""In-tree build of the native pieces (no JIT cache: the .so files travel with the repo snapshot).

  libpgpu.so       hipcc --offload-arch=gfx950   csrc/capi.hip + csrc/host/bignum.cpp   (the product)
  libipcl_amd.so   g++                           csrc/host/*.cpp (ipcl:: C++ API over the C-ABI)
  oracle/*.so      gcc                           oracle/modexp_oracle.c (test infrastructure)
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=cwd)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP extension (there is no CPU fallback)")


def build_pgpu(force=False):
    out = os.path.join(HERE, "libpgpu.so")
    srcs = [os.path.join(CSRC, f) for f in ("capi.hip", "kernels.hpp", "mont_core.hpp", "host/bignum.cpp")]
    srcs += [os.path.join(ROOT, "include", "pgpu.h"), os.path.join(ROOT, "include", "ipcl", "bignum.h")]
    if force or _newer(out, srcs):
        _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
              os.path.join(CSRC, "capi.hip"), os.path.join(CSRC, "host", "bignum.cpp"), "-o", out])
    return out


def build_ipcl(force=False):
    """The ipcl:: C++ host API (mirror of the reference's public classes) on top of libpgpu.so."""
    host = os.path.join(CSRC, "host")
    cpps = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".cpp"))
    if len(cpps) <= 1:
        return None   # only bignum.cpp so far
    out = os.path.join(HERE, "libipcl_amd.so")
    hdrs = []
    for d, _, fs in os.walk(os.path.join(ROOT, "include")):
        hdrs += [os.path.join(d, f) for f in fs]
    if force or _newer(out, cpps + hdrs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-I" + os.path.join(ROOT, "include")]
             + cpps + ["-L" + HERE, "-lpgpu", "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(odir, "libmodexp_oracle.so")
    srcs = [os.path.join(odir, f) for f in ("modexp_oracle.c", "openssl_oracle.c", "ifma_oracle.c", "Makefile")]
    if force or _newer(out, srcs):
        _run(["make", "-C", odir, "-B"])
    return out


def build_all(force=False):
    build_pgpu(force)
    build_ipcl(force)
    build_oracle(force)
