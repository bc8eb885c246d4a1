#!/bin/bash
# evidence of the final build: rocprofv3 passes (tools/profile_r03.sh), default bench line, one-in-flight line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03x
mkdir -p $OUT
cd $REPO
timeout 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
timeout 600 python3 bench.py --in-flight 1 --no-cpu-baseline > $OUT/bench_f1.json 2> $OUT/bench_f1.err; echo "bench f1 rc=$?"
timeout 600 python3 bench.py --config 5 --steps 6 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench c5 rc=$?"
timeout 600 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 rc=$?"
for f in n1 f1 c5 c4; do python3 -c "
import json
d=json.load(open('$OUT/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"; done
bash tools/profile_r03.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"
