#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_periodic.py -q -m gpu -k "through_the_march or pinned_row_on_a_periodic_slab_ring" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log; tail -30 $O/pytest_new.log | cut -c1-250
