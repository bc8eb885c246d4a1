#!/bin/bash
# r03 final: full GPU suite, smoke, the bench line (default), N=2 validation on one device
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03k
mkdir -p $OUT
cd $REPO
timeout 3400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"
BENCH_SINGLE_DEVICE=1 timeout 300 python3 bench.py --gpus 2 --steps 10 --no-extras --no-cpu-baseline > $OUT/bench_n2_pool_1dev.json 2> $OUT/bench_n2_pool_1dev.err; echo "n2 rc=$?"
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_torchrun_1dev.json 2> $OUT/bench_n2_torchrun_1dev.err; echo "n2 torchrun rc=$?"; tail -c 300 $OUT/bench_n2_torchrun_1dev.err
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"
