# Final-build verification, round 6: differential soak, the whole GPU suite, the bench (contract line), smoke
python tools/fuzz_paillier.py 300 6201 2>&1 | tail -1 > gpurun_out/r06_fuzz_final.txt
python tools/fuzz_gpu.py 120 6202 2>&1 | tail -1 >> gpurun_out/r06_fuzz_final.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06_gpu_suite_final3.txt
timeout 900 python bench.py > gpurun_out/r06_bench_final3.line.json 2> gpurun_out/r06_bench_final3.err
cp gpurun_out/bench_detail.json gpurun_out/r06_bench_final3.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke_final3.txt 2>&1
