#!/bin/bash
# tools/abort_hunt2.sh TAG [SEEDS]: the test that has died now and then (test_random_mesh_multigrid_forms_are_bit_identical) on SEEDS
# random meshes instead of 16, behind the parity file as in the runs that died; native frames of a fatal signal in the log (-s)
TAG=${1:-run}; SEEDS=${2:-300}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/abort_hunt; mkdir -p $O
PIB_FUZZ_SEEDS=$SEEDS timeout 1400 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -s -k "parity or multigrid_forms" > $O/$TAG.log 2>&1
rc=$?; echo "rc=$rc" >> $O/$TAG.log
grep -E "passed|failed|Fatal|what\(\)|fatal signal" $O/$TAG.log | tail -4; echo "rc=$rc"
