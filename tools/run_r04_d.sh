#!/bin/bash
# round 4, lease D: the DJN encrypt in the sequential-halves form with a CU claim beside busy lanes (PGPU_ADAPT_ENC_SEQ)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/probe_lanes.py --policy 4 --enc-seq 1 --claim-busy 3 --lanes 2 3 4 --steps 20 2>&1 | grep -v amdgpu
PGPU_ADAPT_ENC_SEQ=1 python tools/probe_trace.py 2 8 2>&1 | grep -v amdgpu | head -28
for nf in 2 4; do for es in 0 1; do
PGPU_ADAPT_ENC_SEQ=$es python bench.py --in-flight $nf --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('in-flight $nf enc_seq $es', d['value'], d['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], d['kernel_forms_in_timed_region'])"
done; done
