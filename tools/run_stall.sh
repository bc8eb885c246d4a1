# Four API threads x CipherText * PlainText: slowest round / slowest operator* call per run (before round 6's fix of
# BaseText::ensureHost every second run had a thread starving 60-350 ms behind another thread's download lock)
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_stall.txt
: > $O
for i in $(seq 1 16); do $B --threads-mul 4 1024 60 2>&1 | grep -v amdgpu.ids >> $O; done
for i in $(seq 1 6); do $B --threads-mul 4 512 60 2>&1 | grep -v amdgpu.ids >> $O; done
for i in $(seq 1 4); do $B --threads-mul 3 1024 60 2>&1 | grep -v amdgpu.ids >> $O; done
for i in $(seq 1 4); do $B --threads 4 700 100 2>&1 | grep -v amdgpu.ids >> $O; done
for i in $(seq 1 4); do $B --threads 4 8192 8 2>&1 | grep -v amdgpu.ids >> $O; done
