#!/bin/bash
# r03: full GPU suite + smoke on the final build, then config 4/5 lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03h
mkdir -p $OUT
cd $REPO
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 3000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 400 python3 bench.py --config 5 --steps 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
timeout 400 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"
BENCH_SINGLE_DEVICE=1 timeout 300 python3 bench.py --gpus 2 --steps 10 --no-extras --no-cpu-baseline > $OUT/bench_n2_pool_1dev.json 2> $OUT/bench_n2_pool_1dev.err; echo "n2 rc=$?"; tail -c 300 $OUT/bench_n2_pool_1dev.err
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_torchrun_1dev.json 2> $OUT/bench_n2_torchrun_1dev.err; echo "n2 torchrun rc=$?"
