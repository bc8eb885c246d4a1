B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_race_sensitivity.txt
: > $O
old=0; new=0
for i in $(seq 1 400); do LD_LIBRARY_PATH=$PWD/pailliercryptolib_amd/oldws:$LD_LIBRARY_PATH PGPU_PLACE_PAD=0 $B --threads-mul 4 1024 2 2>/dev/null | grep -q '"products_ok": true' || old=$((old+1)); done
echo "old workspace code (hipMallocAsync): $old failed runs of 400" >> $O
for i in $(seq 1 400); do PGPU_PLACE_PAD=0 $B --threads-mul 4 1024 2 2>/dev/null | grep -q '"products_ok": true' || new=$((new+1)); done
echo "block-arena workspaces: $new failed runs of 400" >> $O
LD_LIBRARY_PATH=$PWD/pailliercryptolib_amd/oldws:$LD_LIBRARY_PATH ldd $B | grep pgpu >> $O
LD_LIBRARY_PATH=$PWD/pailliercryptolib_amd/oldws:$LD_LIBRARY_PATH python tools/fuzz_threads.py 120 6307 4 2>&1 | tail -1 >> $O
