"""Kernel A/B experiments: builds pailliercryptolib_amd/libpgpu_<name>.so from the regular objects, with the
modexp part(s) recompiled under extra -D flags.  usage: build_variant.py <name> <part,part,..> <flags...>
(part: N = k_modexp.hip part N, hN = k_hensel.hip part N)
Run a benchmark against it with PGPU_LIB=<path>.  (tools/, diagnostics only)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pailliercryptolib_amd import build  # noqa: E402

name, parts, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
if not os.environ.get('PGPU_VARIANT_NO_BASE'):
    build.build_pgpu()
objs = build._objects()
vdir = os.path.join(build.HERE, "build", "variant_" + name)
os.makedirs(vdir, exist_ok=True)
link, jobs = [], []
for o, cmd, _ in objs:
    base = os.path.basename(o)
    num = base.split("_")[-1].split(".")[0]
    part = num if base.startswith("k_modexp_") else "h" + num if base.startswith("k_hensel_") else None
    if part in parts:
        vo = os.path.join(vdir, base)
        jobs.append((vo, cmd))
        link.append(vo)
    else:
        link.append(o)
with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(lambda j: build.compile_one(j[0], j[1], flags), jobs))
out = os.path.join(build.HERE, f"libpgpu_{name}.so")
subprocess.run([build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + link + ["-ldl", "-lpthread", "-o", out],
               check=True)
print(out)
