"""Instruction-alignment pass over the device assembly of the gfx950 kernels (part of the in-tree build).

Measured on MI355X (profiles/r03_code_placement.txt): a wavefront that is ALONE on its SIMD pays ~20 % more for an
8-byte instruction that starts at 4 mod 8 than for one that starts at 0 mod 8 -- and the inner loops of these kernels
are 8-byte VOP3 / DPP instructions almost throughout (v_mad_u64_u32, v_and_b32_dpp, 64-bit shifts and adds).  Every
4-byte instruction in between (a v_and_b32_e32, a v_lshlrev_b32_e32, an s_nop the hazard recogniser inserted) flips the
phase of everything behind it, so which half of a loop runs fast was an accident of code placement: the same source
measured 4.75 or 5.25 ms on the bench's decrypt launch depending on one s_nop in front of the loop.

The pass makes the phase a property of the build:
  * VOP1 / VOP2 / VOPC instructions in their 4-byte `_e32` encoding are re-encoded as `_e64` (same operation, 8 bytes)
    wherever the assembler accepts it (literal operands and a few opcodes have no VOP3 form: those lines are restored);
  * every remaining 4-byte instruction (scalar ALU, s_nop, s_waitcnt, branches) is followed by `.p2align 3`, which the
    assembler fills with one `s_nop 0` where needed -- wait states only ever grow, so hazard padding stays valid.
Nothing else is touched (kernel descriptors, metadata, data sections, s_getpc sequences whose offsets are
position-dependent).  Results are bit-identical by construction and are covered by the whole GPU test suite.

The pass is applied to the kernels with at least MIN_LIMBS limbs per lane (the throughput forms: long runs of 8-byte
instructions between two 4-byte ones, so one padding s_nop buys tens of aligned instructions).  The latency forms
(3-10 limbs per lane) have an s_nop or s_waitcnt every handful of instructions; padding each of them costs more issue
slots than the alignment returns (measured: CRT decrypt of 16 ciphertexts 2.46 -> 2.65 ms with the pass, Encrypt(16)
1.37 -> 1.54 ms), so they are left as hipcc emits them.
"""
import os
import re
import subprocess

_INSTR = re.compile(r"^\t([a-z_0-9]+)(\s|$)")
_E32 = re.compile(r"^\t(v_[a-z_0-9]+)_e32(\s)")
# 4-byte encodings that stay 4 bytes: scalar instructions without a 32-bit literal (the assembler decides; the
# .p2align behind them costs nothing when the position is aligned anyway)
_SCALAR = re.compile(r"^\t(s_[a-z_0-9]+)(\s|$)")
_FUNC = re.compile(r"^(_Z[A-Za-z0-9_]+):")
MIN_LIMBS = int(os.environ.get("PGPU_ALIGN8_MIN_LIMBS", "14"))


def _wanted(mangled):
    """kernels<..., K ...>: the limbs per lane are the second integer template argument (Geo<G,K> or <H,K,...>)"""
    nums = re.findall(r"Li(\d+)E", mangled)
    return len(nums) >= 2 and int(nums[1]) >= MIN_LIMBS


def _rewrite(lines, keep_e32, skip_funcs=()):
    out = []
    in_text = False
    getpc_guard = 0
    active = False
    for i, ln in enumerate(lines):
        if ln.startswith("\t.text") or ln.startswith("\t.section\t.text"):
            in_text = True
        elif ln.startswith("\t.section") or ln.startswith("\t.rodata") or ln.startswith("\t.data") or ln.startswith("\t.amdgpu_metadata"):
            in_text = False
        f = _FUNC.match(ln)
        if f:
            active = _wanted(f.group(1)) and f.group(1) not in skip_funcs
        if not in_text or not active or not _INSTR.match(ln) or ln.startswith("\t."):
            out.append(ln)
            continue
        m = _E32.match(ln)
        if m and i not in keep_e32:
            ln = ln.replace(m.group(1) + "_e32", m.group(1) + "_e64", 1)
            out.append(ln)
            continue
        out.append(ln)
        s = _SCALAR.match(ln)
        if s:
            if s.group(1) == "s_getpc_b64":
                getpc_guard = 4       # the s_add_u32 / s_addc_u32 behind it carry position-dependent offsets
            elif getpc_guard > 0:
                getpc_guard -= 1
            elif s.group(1) not in ("s_endpgm", "s_code_end"):
                out.append("\t.p2align\t3")
        elif m:                       # an _e32 line that had to stay 4 or 8 bytes (literal): realign behind it
            out.append("\t.p2align\t3")
    return out


def align_assembly(src_s, dst_s, assemble_cmd):
    """src_s -> dst_s; assemble_cmd(path) -> (ok, stderr) assembles a candidate.  Lines whose `_e64` form the assembler
    rejects are restored to `_e32`, then the file is assembled once more."""
    lines = open(src_s).read().split("\n")
    keep = set()
    skip = set()
    for _ in range(8):
        cand = _rewrite(lines, keep, skip)
        # map output line numbers back to input lines for error reporting
        open(dst_s, "w").write("\n".join(cand))
        ok, err = assemble_cmd(dst_s)
        if ok:
            return len(keep)
        bad_out = {int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error", err)}
        if not bad_out:
            raise RuntimeError("align8: assembler failed without line information:\n" + err[-2000:])
        # a short branch the compiler had sized for the ORIGINAL code no longer reaches (the pass grows a kernel by a few
        # per cent): leave that kernel as hipcc emitted it (they are the largest ones, which run several wavefronts per
        # SIMD anyway, where placement does not matter)
        grew = {int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error: branch size exceeds", err)}
        if grew:
            for b in grew:
                for k in range(min(b, len(cand)) - 1, -1, -1):
                    f = _FUNC.match(cand[k])
                    if f:
                        skip.add(f.group(1))
                        break
            continue
        # find the input line of each rejected output line: replay the rewrite, tracking indices
        idx_map = _index_map(lines, keep, skip)
        new = {idx_map[b - 1] for b in bad_out if b - 1 in idx_map}
        if not new or new <= keep:
            raise RuntimeError("align8: cannot resolve assembler errors:\n" + err[-2000:])
        keep |= new
    raise RuntimeError("align8: did not converge")


def _index_map(lines, keep_e32, skip_funcs=()):
    """output line index -> input line index, for lines that were re-encoded"""
    out_idx = 0
    mapping = {}
    in_text = False
    getpc_guard = 0
    active = False
    for i, ln in enumerate(lines):
        if ln.startswith("\t.text") or ln.startswith("\t.section\t.text"):
            in_text = True
        elif ln.startswith("\t.section") or ln.startswith("\t.rodata") or ln.startswith("\t.data") or ln.startswith("\t.amdgpu_metadata"):
            in_text = False
        f = _FUNC.match(ln)
        if f:
            active = _wanted(f.group(1)) and f.group(1) not in skip_funcs
        if not in_text or not active or not _INSTR.match(ln) or ln.startswith("\t."):
            out_idx += 1
            continue
        m = _E32.match(ln)
        if m and i not in keep_e32:
            mapping[out_idx] = i
            out_idx += 1
            continue
        out_idx += 1
        s = _SCALAR.match(ln)
        if s:
            if s.group(1) == "s_getpc_b64":
                getpc_guard = 4
            elif getpc_guard > 0:
                getpc_guard -= 1
            elif s.group(1) not in ("s_endpgm", "s_code_end"):
                out_idx += 1
        elif m:
            out_idx += 1
    return mapping


def compile_hip_aligned(hipcc, flags, src, obj, workdir):
    """The hipcc pipeline of one translation unit with the alignment pass between code generation and assembly:
    device assembly -> pass -> assemble -> link (lld) -> bundle -> host object that embeds the bundle."""
    os.makedirs(workdir, exist_ok=True)
    base = os.path.join(workdir, os.path.basename(obj)[:-2])
    llvm = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "clang")):
        llvm = "/opt/rocm/lib/llvm/bin"
    raw_s, fix_s, dev_o, dev_out, fatbin = (base + e for e in (".raw.s", ".s", ".dev.o", ".out", ".hipfb"))
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True, text=True)
    run([hipcc] + flags + ["--cuda-device-only", "-S", src, "-o", raw_s])

    def assemble(path):
        r = subprocess.run([os.path.join(llvm, "clang"), "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", path, "-o", dev_o],
                           capture_output=True, text=True)
        return r.returncode == 0, r.stderr
    kept = align_assembly(raw_s, fix_s, assemble)
    run([os.path.join(llvm, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
    # (--compress: the bundles are zstd/zlib-compressed like hipcc --offload-compress does -- straight-line multiply-
    # accumulate code shrinks 3-4x, the runtime inflates a code object when its module is first used)
    run([os.path.join(llvm, "clang-offload-bundler"), "-type=o", "-bundle-align=4096"] + ([] if os.environ.get("PGPU_NO_COMPRESS") == "1" else ["--compress"]) + [
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + dev_out,
         "-output=" + fatbin])
    run([hipcc] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", src, "-o", obj])
    if os.environ.get("PGPU_ALIGN8_KEEP", "0") == "0":      # the intermediates are ~6 MB per translation unit
        for f in (raw_s, fix_s, dev_o, dev_out, fatbin):
            try:
                os.remove(f)
            except OSError:
                pass
    return kept
