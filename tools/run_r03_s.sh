#!/bin/bash
# validation of the build: full GPU suite, smoke, default bench line, rocprofv3 kernel stats of config 4 (sequential-halves
# decrypt) and of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03s
mkdir -p $OUT
cd $REPO
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$OUT/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['cpu_baseline']['value'])"
cd /tmp && export TMPDIR=/tmp
for cfg in 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c$cfg -- python $REPO/bench.py --config $cfg --steps 4 --warmup 1 > $OUT/trace_c$cfg.log 2>&1
  f=$(ls $OUT/trace_c$cfg/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
done
