#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r06c
for A in "1 0 0 1000" "1 0 1 1000" "1 1 1 1000" "0 1 1 1000" "0 0 1 1000" "1 0 1 -1" "0 0 1 -1"; do
  PIB_PLACE_DEBUG=1 PIB_CRASH_BACKTRACE=1 MALLOC_CHECK_=3 timeout 300 python tools/r06_exit_crash.py $A > gpurun_out/r06c/one.log 2>&1; echo "args [$A] rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06c/one.log | tail -12
done 2>&1 | tee gpurun_out/r06c/exit_crash.txt
