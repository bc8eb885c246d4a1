"""Per-kernel resource usage and per-basic-block instruction mix of a gfx950 assembly file (hipcc -S --cuda-device-only).
usage: asm_stats.py file.s [kernel-substring] [min-block-size]   (tools/, diagnostics only)"""
import collections
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 300
for f in re.split(r'\n\t\.globl\t', s)[1:]:
    name = f.split('\n', 1)[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if want not in dn:
        continue
    g = lambda p: (re.search(p, f) or [None, None])[1]
    print(dn[:90], '| bytes', g(r'; codeLenInByte = (\d+)'), 'vgpr', g(r'; NumVgprs: (\d+)'), 'scratch', g(r'; ScratchSize: (\d+)'),
          'occ', g(r'; Occupancy: (\d+)'), 'lds', g(r'; LDSByteSize: (\d+)'))
    cur = None
    blocks = []
    for ln in f.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            cur = [m.group(1), collections.Counter(), 0]
            blocks.append(cur)
            continue
        t = ln.strip()
        if cur is None or not t or t.startswith(';') or t.startswith('.'):
            continue
        cur[1][t.split()[0]] += 1
        cur[2] += 1
    for b in blocks:
        if b[2] >= minb:
            c = b[1]
            other = {k: v for k, v in c.most_common(14) if k != 'v_mad_u64_u32'}
            print('   ', b[0], 'instr', b[2], 'mad', c['v_mad_u64_u32'], 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
                  'waitcnt', c['s_waitcnt'], 'nop', c['s_nop'], 'scratch', sum(v for k, v in c.items() if 'scratch' in k), other)
