"""Builds a tools/*.hip microbenchmark through the library's compile step (alignment pass included):
usage: build_ubench.py <name>   ->  tools/<name>   (tools/, diagnostics only)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pailliercryptolib_amd import build  # noqa: E402

name = sys.argv[1]
src = os.path.join(ROOT, "tools", name + ".hip")
odir = os.path.join(build.HERE, "build", "variant_ubench")
os.makedirs(odir, exist_ok=True)
obj = os.path.join(odir, name + ".o")
hipcc = build.hipcc_path()
build.compile_one(obj, [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", obj], sys.argv[2:])
subprocess.run([hipcc, "--offload-arch=gfx950", obj, "-o", os.path.join(ROOT, "tools", name)], check=True)
print(os.path.join(ROOT, "tools", name))
