# How many hardware queues do the batch lanes get?  (ROCm maps HIP streams onto GPU_MAX_HW_QUEUES queues, default 4.)
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_threads_queues.txt
: > $O
for q in 4 8 16; do
  for wf in 1 0; do
    for n in 512 1024; do for t in 2 3 4; do echo "GPU_MAX_HW_QUEUES=$q PGPU_WAVE_FORMS=$wf" >> $O; GPU_MAX_HW_QUEUES=$q PGPU_WAVE_FORMS=$wf timeout 120 $B --threads-mul $t $n 60 >> $O 2>&1; done; done
  done
  for n in 700 2048; do for t in 2 4; do echo "GPU_MAX_HW_QUEUES=$q PGPU_WAVE_FORMS=1" >> $O; GPU_MAX_HW_QUEUES=$q timeout 120 $B --threads $t $n 100 >> $O 2>&1; done; done
done
