#!/bin/bash
# alignment pass extended to the forms with 9-10 limbs per lane: decrypt kernel time at mid-size batches
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for lib in libpgpu.so libpgpu_al9.so; do echo $lib; PGPU_LIB=$REPO/pailliercryptolib_amd/$lib timeout 200 python3 tools/bench_decrypt_sizes.py 1024 2048 3000 4096 6000; done
