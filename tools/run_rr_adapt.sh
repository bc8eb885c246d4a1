# Four API threads: the part-chip forms of the adaptive policy (PGPU_RR_ADAPT=3, from 1024 elements per launch) against the
# lone caller's forms (PGPU_RR_ADAPT=0)
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_rr_adapt.txt
: > $O
for rep in 1 2; do
for rr in 3 0; do
  for n in 64 256 700 1024 2048 4096 8192; do for t in 4; do echo "PGPU_RR_ADAPT=$rr" >> $O; PGPU_RR_ADAPT=$rr timeout 120 $B --threads $t $n 100 >> $O 2>&1; done; done
done
done
