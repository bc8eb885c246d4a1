# Four API threads x CipherText * PlainText + a first decrypt each, all at once: the configuration in which 2 % of the runs
# delivered zeros for the tail of a decrypted batch while workspaces grew through hipMallocAsync (profiles/r06_thread_race.txt)
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_bimodal.txt
: > $O
for i in $(seq 1 ${RUNS:-400}); do PGPU_PLACE_PAD=0 $B --threads-mul 4 1024 6 2>&1 | grep -v amdgpu.ids >> $O; done
for i in $(seq 1 100); do $B --threads-mul 4 1024 6 2>&1 | grep -v amdgpu.ids >> $O; done
