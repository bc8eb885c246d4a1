B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_wave_pad.txt
: > $O
for rep in 1 2; do
for pad in 0 192; do
  for n in 64 128 256; do for t in 1 2 3 4; do echo "PGPU_PLACE_PAD=$pad" >> $O; PGPU_PLACE_PAD=$pad timeout 120 $B --threads $t $n 200 >> $O 2>&1; done; done
done
done
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_cpp_api.py -x -q 2>&1 | tail -3 > gpurun_out/r06_wave_pad_tests.txt
