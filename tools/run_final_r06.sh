bash tools/run_place_pad.sh
bash tools/run_threads_small.sh
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06_gpu_suite_final2.txt
timeout 900 python bench.py > gpurun_out/r06_bench_final2.line.json 2> gpurun_out/r06_bench_final2.err
pailliercryptolib_amd/ipcl_api_bench > gpurun_out/r06_ipcl_api_bench_final2.txt 2>&1
