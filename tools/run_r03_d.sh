#!/bin/bash
# r03: aligned build + batch lanes: bench (1 and 2 batches in flight), config 5, config 4, then the whole GPU suite
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], d.get("one_batch_in_flight",{}).get("ms_per_step"), [(k[:26],x.get("ms")) for k,x in r.get("other_kernels",{}).items()])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 python3 bench.py --in-flight 1 --no-extras --no-cpu-baseline > $OUT/bench_f1.json 2> $OUT/bench_f1.err; show $OUT/bench_f1.json
timeout 300 python3 bench.py --in-flight 2 --no-extras --no-cpu-baseline > $OUT/bench_f2.json 2> $OUT/bench_f2.err; show $OUT/bench_f2.json; tail -c 400 $OUT/bench_f2.err
PGPU_PACKED_DECRYPT=0 timeout 300 python3 bench.py --in-flight 1 --no-extras --no-cpu-baseline > $OUT/bench_f1_full.json 2> $OUT/bench_f1_full.err; show $OUT/bench_f1_full.json
timeout 400 python3 bench.py --config 5 --steps 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; show $OUT/bench_c5.json
timeout 400 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; show $OUT/bench_c4.json
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"
timeout 3000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest_gpu.log
