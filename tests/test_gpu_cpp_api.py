"""Builds and runs the C++ test program of the ipcl:: API mirror (tests/cpp/ipcl_api_tests.cpp) on
the GPU: the reference's gtest suites (test_cryptography.cpp, test_ops.cpp) restated against our
drop-in headers.  The binary is compiled here with g++ (host code only; kernels are in libpgpu.so)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
LIBDIR = os.path.join(ROOT, "pailliercryptolib_amd")


def build_test_binary():
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
    names = {"p": "KAT_P", "q": "KAT_Q", "m0": "KAT_M0", "m1": "KAT_M1", "r0": "KAT_R0", "r1": "KAT_R1",
             "c1": "KAT_C1", "c2": "KAT_C2", "c1c2": "KAT_C1C2", "m1m2": "KAT_M1M2",
             "bench_hs": "KAT_BENCH_HS", "bench_r": "KAT_BENCH_R"}
    with open(os.path.join(CPP, "kat_vectors.inc"), "w") as f:
        f.write("// generated from tests/golden/iso_kat.json -- do not edit\n")
        for key, macro in names.items():
            f.write(f'#define {macro} "{k[key]}"\n')
    out = os.path.join(CPP, "ipcl_api_tests.bin")
    for src, exe in (("ipcl_api_tests.cpp", out), ("ipcl_bench.cpp", os.path.join(CPP, "ipcl_bench.bin"))):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + CPP,
                        os.path.join(CPP, src), "-L" + LIBDIR, "-lipcl_amd", "-lpgpu",
                        "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return out


def test_cpp_api_compiles_and_links():
    """CPU-side check: the drop-in headers compile as client code and link against the libraries."""
    from pailliercryptolib_amd import build as b
    b.build_pgpu()
    b.build_ipcl()
    assert os.path.exists(build_test_binary())


@pytest.mark.gpu
def test_cpp_api_suite_on_gpu():
    from pailliercryptolib_amd import build as b
    b.build_pgpu()
    b.build_ipcl()
    exe = build_test_binary()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(r.stdout[-6000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, "C++ API tests failed"
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_api_suite_on_oversubscribed_pool():
    """The same suite on an in-process pool of three entries (wrapped around the box's one GPU) with a tiny minimum
    shard: every batch of the reference's test sizes is cut over three 'devices' -- shard order, broadcast operands,
    per-device key copies and the Montgomery-domain chains all go through the multi-device code."""
    from pailliercryptolib_amd import build as b
    b.build_pgpu()
    b.build_ipcl()
    exe = build_test_binary()
    # (and with the host layer's thread team switched on: IPCL_NUM_THREADS -- off by default, csrc/host/common.cpp)
    env = dict(os.environ, PGPU_POOL_OVERSUBSCRIBE="1", IPCL_GPU_DEVICES="3", PGPU_MIN_SHARD="4", IPCL_EXPECT_POOL="3",
               IPCL_NUM_THREADS="4")
    env.pop("LOCAL_RANK", None)
    env.pop("IPCL_GPU_DEVICE", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-6000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, "C++ API tests failed on the pool"
    assert " 0 failed" in r.stdout
