# Small batches from several host threads, latency forms on (default) and off: encrypt + decrypt round trips and CT x PT
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_threads_small.txt
: > $O
for wf in 1 0; do
  for n in 64 256 700; do for t in 1 2 4; do echo "PGPU_WAVE_FORMS=$wf" >> $O; PGPU_WAVE_FORMS=$wf timeout 120 $B --threads $t $n 300 >> $O 2>&1; done; done
  for n in 128 512 1024; do for t in 1 2 4; do echo "PGPU_WAVE_FORMS=$wf" >> $O; PGPU_WAVE_FORMS=$wf timeout 120 $B --threads-mul $t $n 200 >> $O 2>&1; done; done
done
