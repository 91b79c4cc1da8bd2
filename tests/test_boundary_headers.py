"""The PETSc-facing adapters of the drop-in boundary (include/petibm_amd/{petsc_adapter,linsolver,AmgXSolver}.hpp).

PETSc is not installed in this image, so all this can do is a SYNTAX / type check: `g++ -fsyntax-only` of a translation
unit of this repository (tests/stubs/adapter_syntax_check.cpp) that calls the LinSolverBase mirror and the six
AmgXSolver members with the argument types PetIBM passes (include/petibm/linsolver.h:104-131,
src/linsolver/linsolveramgx.cpp:69-123), against a declarations-only stub of the few PETSc 3.16 / MPI signatures the
adapters use (tests/stubs/petsc).  It pins nothing else -- no behaviour, no parity.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _syntax(args):
    return subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + args,
                          cwd=ROOT, capture_output=True, text=True)


def test_petsc_adapters_parse_against_the_declarations_only_stub():
    r = _syntax(["-I", "tests/stubs/petsc", "-I", "include", "tests/stubs/adapter_syntax_check.cpp"])
    assert r.returncode == 0, r.stderr


def test_amgx_surface_rejects_a_wrong_argument_type():
    """the check has teeth: the same unit with a Vec passed where setA wants a Mat must NOT parse"""
    src = open(os.path.join(ROOT, "tests/stubs/adapter_syntax_check.cpp")).read()
    bad = src.replace("return amgx.setA(A);", "Vec v = nullptr; return amgx.setA(v);")
    assert bad != src
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-I", "tests/stubs/petsc", "-I", "include", "-x", "c++", "-"],
                       cwd=ROOT, input=bad, capture_output=True, text=True)
    assert r.returncode != 0


REFERENCE = "/root/reference"


def _reference_unit(src, extra_inc=()):
    """`g++ -fsyntax-only -DHAVE_AMGX` of one of the reference's OWN source files, read where it lies under /root/reference
    (nothing copied, nothing built), with <AmgXSolver.hpp> resolving to this repository's shim and PETSc / yaml-cpp to the
    declarations-only stubs."""
    inc = []
    for d in list(extra_inc) + ["tests/stubs/petsc", "tests/stubs/yaml", "include/petibm_amd", os.path.join(REFERENCE, "include")]:
        inc += ["-I", d]
    return subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-DHAVE_AMGX"] + inc + [os.path.join(REFERENCE, "src", "linsolver", src)],
                          cwd=ROOT, capture_output=True, text=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "linsolver")), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("src", ["linsolveramgx.cpp", "linsolver.cpp"])
def test_the_references_own_amgx_plug_file_parses_against_the_shim(src):
    """INTEGRATION.md B says src/linsolver/linsolveramgx.cpp compiles UNCHANGED against include/petibm_amd/AmgXSolver.hpp.
    Here the reference's own translation units -- linsolveramgx.cpp (the six AmgXSolver members it calls: initialize, setA,
    solve, getIters, getResidual, finalize; :37-123) and linsolver.cpp (the factory that makes a LinSolverAmgX for `type: GPU`,
    :77-83) -- are type-checked IN PLACE against that shim.  A syntax check, not a build: PETSc and yaml-cpp are
    declarations-only stubs (tests/stubs), nothing is linked or run; skipped where /root/reference does not exist."""
    r = _reference_unit(src)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "linsolver")), reason="the reference tree exists in the build container only")
def test_the_reference_plug_file_check_has_teeth(tmp_path):
    """... and it fails when the shim lacks a member the reference calls (getResidual(iter, res), linsolveramgx.cpp:123)"""
    shim = open(os.path.join(ROOT, "include", "petibm_amd", "AmgXSolver.hpp")).read()
    bad = shim.replace("getResidual", "getResidualRenamed")
    assert bad != shim
    (tmp_path / "AmgXSolver.hpp").write_text(bad)
    r = _reference_unit("linsolveramgx.cpp", extra_inc=[str(tmp_path), os.path.join(ROOT, "include", "petibm_amd")])
    assert r.returncode != 0 and "getResidual" in r.stderr


def test_petsc_ksp_driver_parses_against_the_stub_and_bench_builds_it_only_when_petsc_is_found(monkeypatch, tmp_path):
    """tools/petsc_ksp_driver.c -- the optional genuine-PETSc CPU row of SURVEY.md 8d (KSPCG + GAMG on the bench's own CSR) --
    is SYNTAX-checked against the declarations-only stub (no PETSc in the image: it has never been linked); bench.py builds
    and runs it only when pkg-config / PETSC_DIR resolve, and leaves the line alone otherwise."""
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", "tests/stubs/petsc",
                        "tools/petsc_ksp_driver.c"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("PETSC_DIR", raising=False)
    if bench.petsc_config() is None:  # the build image: nothing found, nothing added
        assert bench.petsc_baseline(16, 1e-10, 1e-3) is None and "port row only" in bench.petsc_probe()
    # a PETSC_DIR that holds no usable library: found, build fails, reported as a note -- never an exception
    fake = tmp_path / "petsc"
    (fake / "include").mkdir(parents=True)
    (fake / "lib").mkdir()
    monkeypatch.setenv("PETSC_DIR", str(fake))
    monkeypatch.setenv("PETSC_ARCH", "")
    if bench.petsc_config() is not None and "PETSC_DIR" in bench.petsc_config()[2]:
        out = bench.petsc_baseline(8, 1e-10, 1e-3)
        assert out["kind"] == "petsc" and out["value"] is None and "did not build" in out["note"]


def test_plain_mirror_still_builds_without_petsc():
    r = _syntax(["-I", "include", "-x", "c++", "-include", "petibm_amd/linsolver.hpp", "/dev/null"])
    assert r.returncode == 0, r.stderr


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-launches itself under torch.distributed.run (2 ranks on
    127.0.0.1).  Without a GPU every rank refuses with ONE JSON line on rank 0 (value null + notes) instead of an assert:
    this exercises the self-launch path on CPU."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["PIB_BENCH_SHARE_GPU"] = "1"  # skip the visible-GPU count check of the launcher itself
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] is None and out["notes"]
