#!/bin/bash
# r02 GPU run C: bench.py in all its modes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02c
mkdir -p $OUT
cd $REPO
t0=$(date +%s)
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$? $(( $(date +%s) - t0 ))s"; tail -c 2500 $OUT/bench_n1.json; tail -3 $OUT/bench_n1.err
BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_pool_1dev.json 2> $OUT/bench_n2_pool_1dev.err; echo "n2 pool rc=$?"; head -c 600 $OUT/bench_n2_pool_1dev.json; tail -3 $OUT/bench_n2_pool_1dev.err
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_torchrun_1dev.json 2> $OUT/bench_n2_torchrun_1dev.err; echo "n2 torchrun rc=$?"; head -c 600 $OUT/bench_n2_torchrun_1dev.json; tail -3 $OUT/bench_n2_torchrun_1dev.err
python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_config4_n1.json 2> $OUT/bench_config4_n1.err; echo "c4 rc=$?"; cat $OUT/bench_config4_n1.json; tail -3 $OUT/bench_config4_n1.err
python3 bench.py --config 5 --steps 8 > $OUT/bench_config5_n1.json 2> $OUT/bench_config5_n1.err; echo "c5 rc=$?"; cat $OUT/bench_config5_n1.json; tail -3 $OUT/bench_config5_n1.err
