# HIP API + kernel trace of four API threads multiplying 1024-element vectors: where does a stalled thread wait?
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/pailliercryptolib_amd/ipcl_api_bench
for i in 1 2 3 4 5 6; do
  rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $R/gpurun_out/thrtrace4_$i -- $B --threads-mul 4 1024 20 2>/dev/null | grep us_per_mul > $R/gpurun_out/thrtrace4_$i.log
done
