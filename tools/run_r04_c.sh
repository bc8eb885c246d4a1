#!/bin/bash
# round 4, lease C: bench line with the lane-activity window and the batches-in-flight sweep
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04c
mkdir -p $OUT
cd $REPO
for nf in 2 4; do
timeout 600 python bench.py --steps 20 --warmup 3 --in-flight $nf --no-cpu-baseline > $OUT/bench_f$nf.json 2> $OUT/bench_f$nf.err
python3 - $OUT/bench_f$nf.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
for k in ("value","ms_per_step","one_batch_in_flight","sustained","kernel_forms_in_timed_region","batches_in_flight_sweep","end_to_end","end_to_end_pinned","end_to_end_two_callers","end_to_end_two_callers_pinned","api_level","extras_error"):
    print(k, d.get(k))
r=d["roofline"]; print(r["kernel"], r["kernel_ms"], r["frac"], r.get("frac_useful")); print(r.get("measured_in_flight"))
PY
done
