#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r03m
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r03m/pytest_durations.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/r03m/pytest_durations.log
