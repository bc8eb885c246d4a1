cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$R/pailliercryptolib_amd/ipcl_api_bench
for wf in 0 1; do
  PGPU_WAVE_FORMS=$wf rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/thrtrace_wf$wf -- $B --threads-mul 2 512 6 > $R/gpurun_out/thrtrace_wf$wf.log 2>&1
done
