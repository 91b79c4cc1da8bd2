#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dmda_boxes.py tests/test_mpi_launch.py tests/test_gpu_structure.py -q -m gpu > $O/pytest_boxes.log 2>&1; echo "pytest rc=$?" >> $O/pytest_boxes.log; tail -4 $O/pytest_boxes.log
PIB_TRACE_SETUP=1 timeout 900 python tools/box_route_probe.py 512 8 0 > $O/box_route_512.txt 2>&1; grep -v amdgpu.ids $O/box_route_512.txt | grep -v "^\[pib setup\] set_csr: \(classify\|upload_csr \|after\|detect\)" | tail -30
