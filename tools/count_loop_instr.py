"""Instructions of the loops of a kernel in a gfx950 assembly file: for every backward branch, the instruction count between
its target label and the branch (nested loops: inner count and whole span).  usage: count_loop_instr.py file.s <kernel-substring>
(tools/, diagnostics only; feeds tools/ubench_ps with instructions per window)"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n\t\.globl\t', text)[1:]:
    name = f.split('\n', 1)[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if want not in dn:
        continue
    lines = f.split('\n')
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', ln)] if m}

    def count(a, b):
        n = mad = 0
        for ln in lines[a:b + 1]:
            t = ln.strip()
            if not t or t[0] in ';.' or t.endswith(':'):
                continue
            n += 1
            mad += t.startswith('v_mad_u64_u32')
        return n, mad
    print(dn[:100])
    for i, ln in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            n, mad = count(labels[m.group(1)], i)
            print(f"   loop {m.group(1)}: {n} instructions, {mad} v_mad_u64_u32 ({100 * mad / n:.1f} %)")
