#!/bin/bash
# r02 evidence run: bench in every mode + rocprofv3 (stats, PMC) + API-level bench; outputs under gpurun_out/r02final
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02final
mkdir -p $OUT
cd $REPO
timeout 300 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"
PGPU_HENSEL=0 timeout 200 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_n1_fullwidth_decrypt.json 2> $OUT/bench_n1_fullwidth_decrypt.err; echo "n1 full-width rc=$?"
PGPU_HENSEL=0 timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench_fullwidth_decrypt.txt 2>&1; echo "api bench full-width rc=$?"
BENCH_SINGLE_DEVICE=1 timeout 200 python3 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_pool_1dev.json 2> $OUT/bench_n2_pool_1dev.err; echo "n2 pool rc=$?"
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 > $OUT/bench_n2_torchrun_1dev.json 2> $OUT/bench_n2_torchrun_1dev.err; echo "n2 torchrun rc=$?"
timeout 300 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_config4_n1.json 2> $OUT/bench_config4_n1.err; echo "c4 rc=$?"
timeout 300 python3 bench.py --config 5 --steps 8 > $OUT/bench_config5_n1.json 2> $OUT/bench_config5_n1.err; echo "c5 rc=$?"
BENCH_SINGLE_DEVICE=1 timeout 300 python3 bench.py --config 5 --gpus 2 --steps 4 > $OUT/bench_config5_n2_pool_1dev.json 2> $OUT/bench_config5_n2_pool_1dev.err; echo "c5 n2 rc=$?"
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"
timeout 600 python3 tools/bench_small_cpu.py > $OUT/small_batch_cpu.txt 2>&1; echo "small cpu rc=$?"
bash tools/profile_r02.sh r02 > $OUT/profile.log 2>&1; echo "profile rc=$?"
