#!/bin/bash
# r02 GPU run B: new tests (pool, streams) + bench line with the restructured library
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02b
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_pool.py tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -40 > $OUT/pytest_new.txt
cat $OUT/pytest_new.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -5 $OUT/bench.err
PGPU_SECRET_EXP=sliding python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_sliding.json 2> $OUT/bench_sliding.err; tail -c 800 $OUT/bench_sliding.json
