#!/bin/bash
# round 6: the three-component march of the explicit terms (k_ns_rhs_march) against the per-component kernels, and two
# register budgets of the march; time-step tests first.  One gpurun call.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_navierstokes.py tests/test_gpu_navierstokes_slabs.py tests/test_gpu_timeintegration.py tests/test_gpu_periodic.py -x -q -m gpu > $O/pytest_ns.log 2>&1; echo "ns tests rc=$?" | tee -a $O/pytest_ns.log; tail -3 $O/pytest_ns.log
# which of the round's new tests leaves a double free at exit
for T in "tests/test_gpu_rccl_selftest.py::test_comm_latency_in_its_own_one_rank_world" "tests/test_gpu_parity.py::test_no_placement_search_on_a_device_whose_memory_is_mostly_taken" "tests/test_gpu_parity.py::test_describe_says_what_runs" "tests/test_gpu_parity.py::test_search_direction_placed_against_x_is_bit_identical" "tests/test_gpu_rccl_selftest.py::test_rccl_entry_points_in_a_one_rank_world"; do
  timeout 300 python -m pytest "$T" -x -q -m gpu > $O/one.log 2>&1; echo "$T rc=$?"; grep -i "double free\|corruption\|Aborted" $O/one.log | head -2
done 2>&1 | tee $O/exit_crash_hunt.txt
for M in 1 0; do
  echo "== PIB_RHS_MARCH=$M (registers for 4 waves per SIMD)"
  PIB_RHS_MARCH=$M timeout 600 python tools/cavity3d_stages.py 256 | tail -2
  PIB_RHS_MARCH=$M timeout 600 python tools/cavity3d_stages.py 512 --steps 6 | tail -2
done 2>&1 | tee $O/stages.txt
bash tools/build_variant.sh w3 navierstokes.hip -DPIB_RHS_WAVES=3 > $O/build_w3.log 2>&1
cp petibm_amd/lib/libpetibm_amd.so /tmp/base.so && cp petibm_amd/lib/var_w3.so petibm_amd/lib/libpetibm_amd.so
echo "== PIB_RHS_MARCH=1 (registers for 3 waves per SIMD)" | tee -a $O/stages.txt
(timeout 600 python tools/cavity3d_stages.py 256 | tail -2; timeout 600 python tools/cavity3d_stages.py 512 --steps 6 | tail -2) 2>&1 | tee -a $O/stages.txt
cp /tmp/base.so petibm_amd/lib/libpetibm_amd.so
