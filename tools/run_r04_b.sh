#!/bin/bash
# round 4, lease B: new GPU tests, adaptive-policy variants on 2..4 lanes, bench line, API bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04b
mkdir -p $OUT
cd $REPO
(time timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cpp_api.py -m gpu -x -q) > $OUT/pytest_new.log 2>&1
tail -15 $OUT/pytest_new.log
timeout 200 python tools/probe_lanes.py --policy 4 --lanes 2 3 4 > $OUT/lanes_a.log 2>&1; cat $OUT/lanes_a.log
timeout 200 python tools/probe_lanes.py --policy 4 --claim-busy 1 --lanes 3 4 > $OUT/lanes_b.log 2>&1; cat $OUT/lanes_b.log
timeout 200 python tools/probe_lanes.py --policy 4 --claim-busy 0 --lanes 2 3 4 > $OUT/lanes_c.log 2>&1; cat $OUT/lanes_c.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.err
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04b/bench.json").read().strip().split("\n")[-1])
for k in ("value","ms_per_step","one_batch_in_flight","sustained","kernel_forms_in_timed_region","end_to_end","end_to_end_pinned","end_to_end_two_callers","api_level","extras_error"):
    print(k, d.get(k))
print(json.dumps(d["roofline"], indent=1)[:4000])
PY
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; cat $OUT/ipcl_api_bench.txt
