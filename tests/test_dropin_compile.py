"""The drop-in claim of INTEGRATION.md, checked: the reference's OWN client sources -- test/*.cpp, benchmark/*.cpp,
example/*.cpp -- parse and type-check UNCHANGED against include/ipcl (our mirror of the public ipcl:: API:
KeyPair / PublicKey / PrivateKey / PlainText / CipherText / BigNumber, ipcl/include/ipcl/*.hpp).

Build container only: the files are compiled where they lie under /root/reference (never copied into this repository);
googletest / google-benchmark are replaced by the two stand-in headers under tests/cpp/shims (syntax only).  Skips when
the reference tree is absent (the GPU box)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SOURCES = sorted(glob.glob(os.path.join(REF, "test", "*.cpp")) + glob.glob(os.path.join(REF, "benchmark", "*.cpp"))
                 + glob.glob(os.path.join(REF, "example", "*.cpp")))


@pytest.mark.skipif(not SOURCES, reason="reference tree not present (it never travels to the GPU box)")
@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("src", SOURCES or ["-"], ids=lambda s: os.path.relpath(s, REF) if s != "-" else s)
def test_reference_client_source_compiles_unchanged(src):
    cmd = ["g++", "-std=c++17", "-fopenmp", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "cpp", "shims"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "%s does not compile against include/ipcl:\n%s" % (os.path.relpath(src, REF), r.stderr[-3000:])


def test_the_check_covers_every_client_source():
    if not SOURCES:
        pytest.skip("reference tree not present")
    assert len(SOURCES) >= 11, SOURCES
