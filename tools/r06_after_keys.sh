#!/bin/bash
# round 6: the GPU suite from the first file that failed after the key consolidation on, then the MPI launch and the box-route probe
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank_loopback.py tests/test_mpi_launch.py -q -m gpu -k 'presmoothing_pair or marching_transfers or residual_update or mpiexec' > $O/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.log; tail -8 $O/pytest_rest.log
(ls /opt/conda/bin/mpiexec && for P in 1 2 4 8; do PIB_TRANSPORT=peer timeout 300 /opt/conda/bin/mpiexec -n $P examples/mpi/poisson_boxes_mpi $((P == 8 ? 64 : 48)) 1e-10 1; echo "mpiexec -n $P rc=$?"; done) > $O/mpiexec_boxes.txt 2>&1; grep -v amdgpu.ids $O/mpiexec_boxes.txt | tail -14
