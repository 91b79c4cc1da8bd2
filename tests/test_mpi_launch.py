"""The launch model of an unchanged PetIBM -- `mpiexec -n P`, one process per GPU, rows in PETSC_DECIDE's boxes -- through the C
ABI with REAL MPI calls and no PETSc (examples/mpi/poisson_boxes_mpi.cpp: what include/petibm_amd/petsc_adapter.hpp does with
PETSc's communicator, src/linsolver/linsolveramgx.cpp:69,84,96).  MPICH 3.3.2 is part of the image; skipped where it is not."""
import json
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.fixture(scope="module")
def mpi_binary(built_library):
    from petibm_amd.build import build_mpi_example, mpiexec_path
    exe = build_mpi_example()
    if exe is None or mpiexec_path() is None:
        pytest.skip("no MPI in this image")
    return exe, mpiexec_path()


def test_mpi_example_compiles_links_and_fails_loudly_without_a_gpu(mpi_binary):
    """plain g++ + mpi.h + libpetibm_amd.so: the binary resolves the library and the MPI runtime (and NOT the old libstdc++ that
    sits beside the MPI runtime); without a GPU the first library call fails with a message and MPI_Abort, never silently."""
    exe, mpiexec = mpi_binary
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpetibm_amd.so" in ldd and "libmpi.so.12" in ldd and "not found" not in ldd
    stdcpp = [ln for ln in ldd.splitlines() if "libstdc++" in ln]
    assert stdcpp and "conda" not in stdcpp[0]
    if os.path.exists("/dev/kfd"):  # (no torch in this process: tests/conftest.py)
        pytest.skip("a GPU is present: the run itself is tests/test_mpi_launch.py::test_mpiexec_ranks_on_petsc_decide_boxes")
    out = subprocess.run([mpiexec, "-n", "2", exe, "8"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "failed with" in out.stderr + out.stdout or "no ROCm-capable device" in out.stderr + out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,n", [(1, 32), (2, 32), (4, 32), (8, 48)])
def test_mpiexec_ranks_on_petsc_decide_boxes(mpi_binary, ranks, n):
    """`PIB_TRANSPORT=peer mpiexec -n P poisson_boxes_mpi N`: the id through MPI_Bcast, the device from the node-local rank, every
    rank's (m, n, p) box through pib_set_csr_i32 only, multigrid-PCG to 1e-10 -- residual contract, error against the
    manufactured solution and the single-rank iteration count on 1 / 2 / 4 / 8 processes sharing the GPU (RCCL refuses several
    ranks per device; on a node with P GPUs the same line runs over RCCL without the variable)."""
    exe, mpiexec = mpi_binary
    env = dict(os.environ, PIB_TRANSPORT="peer", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([mpiexec, "-n", str(ranks), exe, str(n), "1e-10", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    grids = {1: [1, 1, 1], 2: [1, 1, 2], 4: [1, 2, 2], 8: [2, 2, 2]}
    assert d["ok"] and d["ranks"] == ranks and d["grid"] == grids[ranks]
    assert d["true_rel_residual"] <= 1e-9 and d["max_error"] <= 1e-6
    assert 5 <= d["iters"] <= 14
    runs = [ln for ln in out.stdout.splitlines() if ln.startswith("runs:")][-1]
    assert f"ranks={ranks}" in runs and ("partition=boxes_to_slabs" in runs if ranks >= 4 else True)
