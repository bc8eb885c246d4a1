O=gpurun_out/r06_torchrun_cmp.txt
: > $O
run() { echo "== $1" >> $O; shift; "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" >> $O; }
run "direct steps 10" python bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline
run "direct steps 20" python bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline
run "torchrun steps 10" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline
run "torchrun steps 20" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline
export OMP_NUM_THREADS=16
run "torchrun steps 20 OMP_NUM_THREADS=16" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline
