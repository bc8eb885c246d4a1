#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02d
mkdir -p $OUT
cd $REPO
bash tools/run_variants.sh base prio0 2>&1 | tee $OUT/variants.txt
cat > /tmp/q.py <<'P'
import sys
sys.path.insert(0, ".")
sys.argv=["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("qb", "tools/quick_bench.py")
src = open("tools/quick_bench.py").read().split("pa.initialize()")[0]
exec(src)
pa.initialize()
for n in (16384, 32768, 65536):
    run(1024, 512, n, True)
    run(1024, 1024, n, False)
P
for g in 1 0; do echo "PGPU_GEO_410=$g"; PGPU_GEO_410=$g python /tmp/q.py; done 2>&1 | grep -v amdgpu.ids | tee $OUT/geo410.txt
