"""In-tree build of the native pieces (no JIT cache: the .so files travel with the repo snapshot).

  libpgpu.so       hipcc --offload-arch=gfx950   csrc/k_*.hip (kernels) + g++ csrc/capi.cpp, runtime.cpp,
                                                 host/bignum.cpp                        (the product)
  libipcl_amd.so   g++                           csrc/host/*.cpp (ipcl:: C++ API over the C-ABI)
  oracle/*.so      gcc                           oracle/modexp_oracle.c (test infrastructure)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), file=sys.stderr, flush=True)   # (stdout belongs to the caller: bench.py's JSON line)
    subprocess.run(cmd, check=True, cwd=cwd)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP extension (there is no CPU fallback)")


def build_4096():
    """PGPU_BUILD_4096=1 also builds the split forms of the 4096-bit key class (k_hensel.hip parts 22-24 and the (4,18) /
    (8,9) decrypt forms): beyond every BASELINE config and the reference's own 2048-bit cap (ipcl/keygen.cpp:10), and
    10-15 minutes of compile time per translation unit.  Off by default: such keys take the full-width kernels."""
    return os.environ.get("PGPU_BUILD_4096", "0") == "1"


def _switches():
    return [f"-DPGPU_WITH_4096={1 if build_4096() else 0}"]


def hensel_parts():
    """translation units of k_hensel.hip that the current switches ask for"""
    skip = {15, 30}      # (retired in round 6: the A/B-wavefront experiment and the operand-scanning one-lane kernel)
    if not build_4096():
        skip |= {22, 23, 24}
    return [p for p in range(38) if p not in skip]


def _objects():
    """(object file, compile command, dependencies) of every translation unit of libpgpu.so.  The device code
    is split so that it compiles in parallel (k_modexp.hip once per PGPU_PART, k_misc.hip) and a change of the
    host runtime never recompiles a kernel."""
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    hip = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + _switches() + inc
    host = ["g++", "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + _switches() + inc
    kdeps = [os.path.join(CSRC, f) for f in ("kernels.hpp", "mont_core.hpp", "kargs.hpp", "launch.hpp")]
    hdeps = [os.path.join(CSRC, f) for f in ("kargs.hpp", "launch.hpp", "runtime.hpp", "policy.hpp")]
    hdeps += [os.path.join(ROOT, "include", "pgpu.h"), os.path.join(ROOT, "include", "ipcl", "bignum.h"),
              os.path.join(ROOT, "include", "ipcl", "utils", "serialize.hpp")]
    obj = os.path.join(HERE, "build")
    out = []
    for part in range(8):
        src = os.path.join(CSRC, "k_modexp.hip")
        o = os.path.join(obj, f"k_modexp_{part}.o")
        out.append((o, hip + [f"-DPGPU_PART={part}", "-c", src, "-o", o], [src] + kdeps))
    src = os.path.join(CSRC, "k_misc.hip")
    out.append((os.path.join(obj, "k_misc.o"), hip + ["-c", src, "-o", os.path.join(obj, "k_misc.o")], [src] + kdeps))
    for part in hensel_parts():
        src = os.path.join(CSRC, "k_hensel.hip")
        o = os.path.join(obj, f"k_hensel_{part}.o")
        hdeps_k = [os.path.join(CSRC, f) for f in ("hensel.hpp", "hensel_q.hpp", "hensel_seq.hpp")]
        if part in (31, 33, 34, 35, 36, 37):
            hdeps_k.append(os.path.join(CSRC, "hensel_ps.hpp"))
        if part in (35, 36, 37):
            hdeps_k.append(os.path.join(CSRC, "hensel_wave.hpp"))
        if part in (36, 37):
            hdeps_k.append(os.path.join(CSRC, "hensel_wave_n2.hpp"))
        out.append((o, hip + [f"-DPGPU_PART={part}", "-c", src, "-o", o], [src] + hdeps_k + kdeps))
    for name in ("capi.cpp", "policy.cpp", "runtime.cpp", os.path.join("host", "bignum.cpp")):
        src = os.path.join(CSRC, name)
        o = os.path.join(obj, os.path.basename(name).replace(".cpp", ".o"))
        extra = [os.path.join(CSRC, f) for f in ("capi_keys.inc", "capi_batches.inc", "capi_timing.inc")] if name == "capi.cpp" else []
        out.append((o, host + ["-c", src, "-o", o], [src] + hdeps + extra))
    return out


def align8_enabled():
    """PGPU_ALIGN8=0 builds the device code exactly as hipcc emits it (A/B of the alignment pass, align8.py)"""
    return os.environ.get("PGPU_ALIGN8", "1") != "0"


def compile_one(obj, cmd, extra_flags=()):
    """One translation unit.  Device code (.hip) goes through the instruction-alignment pass (align8.py) unless
    PGPU_ALIGN8=0; host code is a plain compiler call."""
    cmd = list(cmd)
    src = cmd[cmd.index("-c") + 1]
    if src.endswith(".hip") and align8_enabled():
        try:
            from . import align8
        except ImportError:      # imported as a top-level module (cwd = the package directory)
            import align8
        flags = [c for c in cmd[1:] if c not in ("-c", src, "-o", cmd[cmd.index("-o") + 1])] + list(extra_flags)
        print("[build] (aligned)", " ".join([cmd[0]] + flags + ["-c", src, "-o", obj]), file=sys.stderr, flush=True)
        # The pass re-creates hipcc's device pipeline by hand (clang -> lld -> clang-offload-bundler under
        # /opt/rocm/lib/llvm/bin, ROCm 7.2 layout) and rewrites assembly text.  On another ROCm layout, or when the
        # assembler rejects the rewrite, the translation unit is compiled by plain hipcc instead: the result is the same
        # code without the placement guarantee (a lone wavefront then runs 0-10 % slower, profiles/r03_code_placement.txt).
        # Which path an object took is recorded beside it (<object>.how) and printed.
        try:
            align8.compile_hip_aligned(cmd[0], flags, src, obj, os.path.join(os.path.dirname(obj), "align8"))
            _note_how(obj, "aligned")
            return
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:
            detail = getattr(e, "stderr", None) or str(e)
            print(f"[build] WARNING: alignment pass failed for {os.path.basename(obj)} ({str(detail).strip()[-300:]}); "
                  "falling back to plain hipcc", file=sys.stderr, flush=True)
            _note_how(obj, "plain (alignment pass failed)")
    else:
        _note_how(obj, "plain")
    cmd[cmd.index("-o") + 1] = obj
    if src.endswith(".hip") and os.environ.get("PGPU_NO_COMPRESS") != "1":
        cmd = cmd + ["--offload-compress"]      # (the aligned pipeline passes --compress to the bundler itself)
    _run(cmd + list(extra_flags))


def _note_how(obj, how):
    try:
        open(obj + ".how", "w").write(how + "\n")
    except OSError:
        pass


def build_pgpu(force=False):
    from concurrent.futures import ThreadPoolExecutor
    out = os.path.join(HERE, "libpgpu.so")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs = _objects()
    stamp = os.path.join(HERE, "build", "align8.on" if align8_enabled() else "align8.off")
    if not os.path.exists(stamp):      # the pass was switched: every device object is stale
        for f in ("align8.on", "align8.off"):
            if os.path.exists(os.path.join(HERE, "build", f)):
                os.remove(os.path.join(HERE, "build", f))
        force_dev = True
    else:
        force_dev = False
    # the build switches change what launch.hpp declares as compiled: every object that includes it is stale when they
    # change (the k_hensel parts that hold gated forms, and the host side)
    cfg = " ".join(_switches())
    cfg_path = os.path.join(HERE, "build", "switches.txt")
    old_cfg = open(cfg_path).read().strip() if os.path.exists(cfg_path) else None
    cfg_changed = old_cfg is not None and old_cfg != cfg
    gated = ("k_hensel_1.o", "k_hensel_2.o", "capi.o")
    kdep = os.path.join(HERE, "align8.py")
    todo = [(o, cmd) for o, cmd, deps in objs
            if force or _newer(o, deps + ([kdep] if cmd[-3].endswith(".hip") else [])) or (force_dev and cmd[-3].endswith(".hip"))
            or (cfg_changed and os.path.basename(o) in gated)]
    if todo:
        # the 8-lane x 18-limb forms (4096-bit key class, parts 22-24; PGPU_BUILD_4096=1) compile for 10-15 minutes each:
        # start them first; so are the one-lane product-scanning forms (parts 33, 31: fully unrolled column loops, 5 and 3 minutes)
        slow = ("k_hensel_22.", "k_hensel_23.", "k_hensel_24.", "k_hensel_33.", "k_hensel_31.", "k_hensel_37.", "k_hensel_36.", "k_hensel_35.")
        todo.sort(key=lambda oc: 0 if os.path.basename(oc[0]).startswith(slow) else 1)
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda oc: compile_one(oc[0], oc[1]), todo))
    open(stamp, "w").close()
    open(cfg_path, "w").write(cfg + "\n")
    relink = cfg_changed or old_cfg is None
    if todo or relink or not os.path.exists(out):
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _, _ in objs]
             + ["-ldl", "-lpthread", "-o", out])
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


def build_api_bench(force=False):
    """tests/cpp/ipcl_bench.cpp -> pailliercryptolib_amd/ipcl_api_bench: the reference's google-benchmark cases restated
    over the ipcl:: mirror; bench.py runs it with --json for the API-level number."""
    import json
    cpp = os.path.join(ROOT, "tests", "cpp")
    inc = os.path.join(cpp, "kat_vectors.inc")
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
    names = {"p": "KAT_P", "q": "KAT_Q", "m0": "KAT_M0", "m1": "KAT_M1", "r0": "KAT_R0", "r1": "KAT_R1",
             "c1": "KAT_C1", "c2": "KAT_C2", "c1c2": "KAT_C1C2", "m1m2": "KAT_M1M2",
             "bench_hs": "KAT_BENCH_HS", "bench_r": "KAT_BENCH_R"}
    text = "// generated from tests/golden/iso_kat.json -- do not edit\n" + "".join(
        f'#define {macro} "{k[key]}"\n' for key, macro in names.items())
    if not os.path.exists(inc) or open(inc).read() != text:
        open(inc, "w").write(text)
    out = os.path.join(HERE, "ipcl_api_bench")
    src = os.path.join(cpp, "ipcl_bench.cpp")
    libs = [os.path.join(HERE, "libipcl_amd.so"), os.path.join(HERE, "libpgpu.so")]
    build_pgpu()
    build_ipcl()
    if force or _newer(out, [src, inc] + libs):
        _run(["g++", "-O2", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + cpp, src,
              "-L" + HERE, "-lipcl_amd", "-lpgpu", "-Wl,-rpath,$ORIGIN", "-o", out])
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
    build_api_bench(force)
    build_oracle(force)
