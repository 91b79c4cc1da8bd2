#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r06c
for K in "test_search_direction_placed_against_x_is_bit_identical[0]" "test_search_direction_placed_against_x_is_bit_identical[1]"; do
  PIB_PLACE_DEBUG=0 MALLOC_CHECK_=3 timeout 300 python -X faulthandler -m pytest "tests/test_gpu_parity.py::$K" -x -q -s -m gpu > gpurun_out/r06c/one.log 2>&1; echo "[$K] rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06c/one.log | tail -40
done 2>&1 | tee gpurun_out/r06c/exit_crash2.txt
