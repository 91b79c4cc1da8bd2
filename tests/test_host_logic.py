"""CPU tests of the host logic and of the C-ABI surface (no GPU, no compute calls)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The keys of the AmgX-style files the reference ships for its GPU examples
# (examples/navierstokes/liddrivencavity2dRe1000_GPU/config/poisson_solver.info) -- own text, same keys.
AMGX_POISSON = """
config_version=2
communicator=MPI
determinism_flag=1
solver(solv)=PCG
solv:max_iters=1000
solv:monitor_residual=1
solv:convergence=ABSOLUTE
solv:tolerance=1.0E-06
solv:norm=L2
solv:store_res_history=1
solv:preconditioner(prec)=AMG
prec:algorithm=CLASSICAL
prec:max_iters=1
prec:cycle=V
prec:presweeps=1
prec:postsweeps=1
prec:max_levels=100
prec:min_coarse_rows=2
prec:coarse_solver(c_solver)=DENSE_LU_SOLVER
prec:dense_lu_num_rows=128
prec:coarsest_sweeps=1
prec:smoother(smooth)=BLOCK_JACOBI
smooth:relaxation_factor=0.9
"""
AMGX_VELOCITY = """
config_version=2
solver(solv)=PBICGSTAB
solv:max_iters=1000
solv:monitor_residual=1
solv:convergence=ABSOLUTE
solv:tolerance=1.0E-14
solv:norm=L2
solv:store_res_history=1
solv:preconditioner(prec)=BLOCK_JACOBI
prec:relaxation_factor=0.9
prec:max_iters=1
"""
PETSC_BOTH = """
# Poisson solver: prefix `-poisson_`
-poisson_ksp_type cg
-poisson_ksp_atol 1.0E-06
-poisson_ksp_rtol 0.0
-poisson_ksp_max_it 1000
-poisson_pc_type gamg
-poisson_pc_gamg_type agg
-poisson_pc_gamg_agg_nsmooths 1
# velocity solver: prefix `-velocity_`
-velocity_ksp_type bcgs
-velocity_ksp_atol 1.0E-06
-velocity_ksp_rtol 0.0
-velocity_ksp_max_it 1000
-velocity_pc_type jacobi
-velocity_pc_jacobi_type diagonal
"""


@pytest.fixture(scope="module")
def capi(built_library):
    from petibm_amd import capi
    capi.load()
    return capi


def test_library_exports_every_declared_symbol(built_library):
    """Every function include/petibm_amd.h declares is exported by the built library and bound in capi.py."""
    header = open(os.path.join(ROOT, "include", "petibm_amd.h")).read()
    declared = set(re.findall(r"\b(pib_[a-z0-9_]+)\s*\(", header))
    from petibm_amd import capi
    syms = subprocess.check_output(["nm", "-D", "--defined-only", built_library], text=True)
    exported = {ln.split()[-1] for ln in syms.splitlines() if ln.strip()}
    assert declared <= exported, declared - exported
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    # C ABI: no torch / C++ types leak into the exported names
    assert all(not s.startswith("_Z") for s in declared)
    # the header is plain C
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "petibm_amd.h")])


def test_library_loads_and_has_no_cpu_fallback(capi):
    lib = capi.load()
    assert lib.pib_version() >= 100
    # (no `import torch` here: with the library already on /opt/rocm's runtime -- PIB_TORCH_FIRST=0, tests/conftest.py -- torch would
    # bring the copies it bundles into the same process, and the two runtimes' exit handlers trip over each other)
    h = ctypes.c_void_p()
    code = lib.pib_create_from_string(ctypes.byref(h), b"poisson", b"", 0, 1, None, -1)
    if code == 0:  # a GPU box
        assert lib.pib_destroy(h) == 0
    else:
        assert code == capi.ERR_LIB and b"no CPU fallback" in lib.pib_last_error()


def test_peer_transport_ids_are_fresh_shared_memory_names(capi):
    """pib_comm_peer_id needs no GPU: a magic + the name of the POSIX shared-memory segment the ranks will meet in, never the
    same twice; a null output is an error like everywhere in the C ABI."""
    lib = capi.load()
    ids = []
    for _ in range(3):
        buf = ctypes.create_string_buffer(capi.UID_BYTES)
        assert lib.pib_comm_peer_id(buf) == 0
        assert buf.raw[:8] == b"PIBPEER1"
        name = buf.raw[8:].split(b"\0", 1)[0]
        assert name.startswith(b"/pib_peerD_") and b"/" not in name[1:] and len(name) < 100  # D: device-ordered collectives (the default)
        ids.append(name)
    assert len(set(ids)) == 3
    buf = ctypes.create_string_buffer(capi.UID_BYTES)
    assert lib.pib_comm_peer_id_ordered(buf, 0) == 0 and buf.raw[8:].startswith(b"/pib_peerH_")  # host-ordered, on request
    assert lib.pib_comm_peer_id(None) != 0 and b"null output" in lib.pib_last_error()


def test_amgx_config_subset(capi):
    d = capi.config_describe("poisson", AMGX_POISSON)
    assert d["flavor"] == "amgx" and d["type"] == "NVIDIA AmgX"
    assert d["method"] == "cg" and d["pc"] == "gmg" and d["norm"] == "unpreconditioned"
    assert int(d["max_iters"]) == 1000 and float(d["atol"]) == 1e-6 and float(d["rtol"]) == 0.0
    assert d["smoother"] == "jacobi" and float(d["smoother_relaxation"]) == 0.9
    assert (int(d["presweeps"]), int(d["postsweeps"])) == (1, 1) and int(d["guess_nonzero"]) == 1
    v = capi.config_describe("velocity", AMGX_VELOCITY)
    assert v["method"] == "bicgstab" and v["pc"] == "jacobi" and float(v["jacobi_relaxation"]) == 0.9
    assert float(v["atol"]) == 1e-14
    # an empty file = AmgX defaults (linsolveramgx.cpp:62-72 writes an empty temporary file)
    e = capi.config_describe("forces", "")
    assert e["method"] == "cg" and e["pc"] == "none" and int(e["max_iters"]) == 100 and int(e["monitor"]) == 0
    r = capi.config_describe("poisson", AMGX_POISSON.replace("ABSOLUTE", "RELATIVE_INI_CORE"))
    assert float(r["rtol"]) == 1e-6 and float(r["atol"]) == 0.0
    # keys without a scope fall back to the default scope
    f = capi.config_describe("p", "solver=PCG\nmax_iters=7\ntolerance=1e-3\nmonitor_residual=1\n")
    assert int(f["max_iters"]) == 7 and float(f["atol"]) == 1e-3


def test_petsc_options_subset(capi):
    p = capi.config_describe("poisson", PETSC_BOTH)
    assert p["flavor"] == "ksp" and p["type"] == "PETSc KSP"
    assert p["method"] == "cg" and p["pc"] == "gmg" and p["norm"] == "preconditioned"
    assert float(p["atol"]) == 1e-6 and float(p["rtol"]) == 0.0 and int(p["max_iters"]) == 1000
    assert int(p["guess_nonzero"]) == 0 and int(p["error_if_not_converged"]) == 1 and float(p["dtol"]) == 1e4
    v = capi.config_describe("velocity", PETSC_BOTH)
    assert v["method"] == "bicgstab" and v["pc"] == "jacobi"
    # another solver's prefix does not leak: KSP defaults (rtol 1e-5, atol 1e-50, 10000 its, CG)
    f = capi.config_describe("forces", PETSC_BOTH)
    assert f["method"] == "cg" and float(f["rtol"]) == 1e-5 and int(f["max_iters"]) == 10000
    # every forces_solver.info of the reference's decoupled-IBPM examples: a direct solve
    # (examples/decoupledibpm/cylinder2dRe40_GPU/config/forces_solver.info)
    d = capi.config_describe("forces", "# forces solver: prefix `-forces_`\n-forces_ksp_type preonly\n-forces_pc_type lu\n"
                                       "-forces_pc_factor_mat_solver_type superlu_dist\n")
    assert d["method"] == "preonly" and d["pc"] == "lu"
    a = capi.config_describe("forces", "config_version=2\nsolver(s)=DENSE_LU_SOLVER\n")
    assert a["method"] == "preonly" and a["pc"] == "lu" and a["flavor"] == "amgx"
    # PETSc's single-reduction CG (KSPCGUseSingleReduction) by its own option name, per prefix; off by default; the AmgX-style key
    assert int(p["cg_single_reduction"]) == 0
    sr = capi.config_describe("poisson", PETSC_BOTH + "-poisson_ksp_cg_single_reduction\n")
    assert int(sr["cg_single_reduction"]) == 1 and sr["method"] == "cg"
    assert int(capi.config_describe("velocity", PETSC_BOTH + "-poisson_ksp_cg_single_reduction true\n")["cg_single_reduction"]) == 0
    assert int(capi.config_describe("poisson", AMGX_POISSON + "pib_cg_single_reduction=1\n")["cg_single_reduction"]) == 1


@pytest.mark.parametrize("text,code", [
    ("solver(s)=FGMRES\n", 56), ("solver(s)=PCG\ns:preconditioner(p)=MULTICOLOR_DILU\n", 56),
    ("solver(s)=PCG\ns:convergence=RELATIVE_MAX\n", 56), ("solver(s)=PCG\ns:norm=L1\n", 56),
    ("this is not a config\n", 62), ("-poisson_ksp_type gmres\n", 56), ("-poisson_pc_type lu\n", 56),
    ("-poisson_ksp_type preonly\n-poisson_pc_type jacobi\n", 56), ("-poisson_pc_type ilu\n", 56),
])
def test_unsupported_config_is_an_error(capi, text, code):
    with pytest.raises(capi.PibError) as ei:
        capi.config_describe("poisson", text)
    assert ei.value.code == code


def test_missing_config_file_and_null_arguments(capi):
    lib = capi.load()
    h = ctypes.c_void_p()
    assert lib.pib_create(ctypes.byref(h), b"poisson", b"/nonexistent/solver.info", 0, 1, None, -1) == capi.ERR_FILE_OPEN
    assert lib.pib_create(None, b"poisson", None, 0, 1, None, -1) in (capi.ERR_ARG_NULL, capi.ERR_LIB)
    assert lib.pib_get_iters(None, None) == capi.ERR_ARG_NULL
    assert lib.pib_solve(None, None, None) == capi.ERR_ARG_NULL
    assert lib.pib_destroy(None) == 0


def test_createlinsolver_factory_dispatch(capi, tmp_path):
    """src/linsolver/linsolver.cpp:57-91: default type CPU, relative config path joined to `directory`,
    unknown type -> PETSC_ERR_ARG_WRONG (62)."""
    from petibm_amd import linsolver
    node = {"directory": str(tmp_path), "parameters": {"poissonSolver": {"type": "GPU", "config": "config/p.info"}}}
    with pytest.raises(capi.PibError) as ei:
        linsolver.createLinSolver("poisson", node)  # file does not exist -> the joined path is reported
    assert ei.value.code == capi.ERR_FILE_OPEN and str(tmp_path / "config" / "p.info") in ei.value.message
    with pytest.raises(capi.PibError) as ei:
        linsolver.createLinSolver("velocity", {"parameters": {}})  # default type: CPU, not provided here
    assert ei.value.code == capi.ERR_ARG_WRONG
    with pytest.raises(capi.PibError) as ei:
        linsolver.createLinSolver("poisson", {"parameters": {"poissonSolver": {"type": "TPU"}}})
    assert ei.value.code == capi.ERR_ARG_WRONG and "Unrecognized value" in ei.value.message


def test_slab_range_matches_python_partition_and_dmda_rule(capi):
    import slab_plans as partition
    from oracle import mesh as omesh
    lib = capi.load()
    for n, P in ((512, 8), (512, 4), (256, 8), (10, 4), (7, 3), (450, 8), (5, 5)):
        got = []
        for r in range(P):
            b, e = ctypes.c_int64(), ctypes.c_int64()
            assert lib.pib_slab_range(n, P, r, ctypes.byref(b), ctypes.byref(e)) == 0
            got.append((b.value, e.value))
            assert partition.slab_range(n, P, r) == (b.value, e.value)
        assert got == omesh.slab_ranges(n, P)
        assert got[0][0] == 0 and got[-1][1] == n and all(got[i][1] == got[i + 1][0] for i in range(P - 1))
    plans = partition.all_plans((512, 512, 512), 8)
    assert all(p.n_local == 64 * 512 * 512 for p in plans)
    assert plans[0].ghost_lo == 0 and plans[0].ghost_hi == 512 * 512 and plans[7].ghost_hi == 0
    # what a rank sends is what its neighbour receives
    for a, b in zip(plans[:-1], plans[1:]):
        assert a.send_next == b.ghost_lo and b.send_prev == a.ghost_hi


def test_every_solver_file_key_of_the_backend_is_documented():
    """INTEGRATION.md's key table lists every `pib_*` key csrc/config.cpp reads (and nothing it does not read)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = open(os.path.join(root, "petibm_amd", "csrc", "config.cpp")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    read = set(re.findall(r'get\("(pib_[a-z0-9_]+)"', cfg)) | set(re.findall(r'"default", "(pib_[a-z0-9_]+)"', cfg))
    table = "\n".join(line for line in doc.splitlines() if line.startswith("| `pib_"))
    listed = {k for k in re.findall(r"`(pib_[a-z0-9_]+)`", table) if not k.startswith(("pib_ns_", "pib_get_", "pib_set_", "pib_assemble",
                                                                                      "pib_create", "pib_solve", "pib_comm", "pib_slab",
                                                                                      "pib_mat_", "pib_destroy", "pib_time", "pib_memcpy",
                                                                                      "pib_config", "pib_last", "pib_synchronize", "pib_version"))}
    assert read - listed == set(), f"undocumented keys: {sorted(read - listed)}"
    assert listed - read == set(), f"documented but not read: {sorted(listed - read)}"


def test_an_exception_inside_the_library_comes_back_as_an_error_code(built_library, tmp_path):
    """Every extern "C" entry point is a function-try-block (csrc/config.cpp: fail_exception): a C++ exception must not unwind
    into the C / ctypes / cgo caller (std::terminate: the host application aborted by its linear solver).  The reference's
    own boundary behaves like this: PETSc functions return error codes (CHKERRQ, /root/reference/src/linsolver/linsolverksp.cpp:48-60).
    The throw is drilled in a TEST VARIANT of the library -- csrc/capi.cpp recompiled with -DPIB_TEST_HOOKS and linked with the
    product's other objects, loaded in a process of its own; the product library carries no test hook (and ignores the variable)."""
    from petibm_amd import build as B
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(os.path.dirname(B.library_path()), "obj")
    hook_o, hook_so = str(tmp_path / "capi_hooks.o"), str(tmp_path / "libpetibm_amd_hooks.so")
    cflags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
    subprocess.check_call([hipcc] + cflags + ["-DPIB_TEST_HOOKS", "-x", "hip", "-c", os.path.join(ROOT, "petibm_amd", "csrc", "capi.cpp"), "-o", hook_o])
    objs = [hook_o] + [os.path.join(objdir, os.path.splitext(f)[0] + ".o") for f in B.SOURCES if f != "capi.cpp"]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", hook_so] + objs + ["-lrccl"])
    code = ("import ctypes as C, os, sys\n"
            "lib = C.CDLL(sys.argv[1])\n"
            "lib.pib_last_error.restype = C.c_char_p\n"
            "buf = C.create_string_buffer(2048)\n"
            "for val in ('drill', 'm', None):\n"
            "    if val is None: os.environ.pop('PIB_TEST_THROW')\n"
            "    else: os.environ['PIB_TEST_THROW'] = val\n"
            "    rc = lib.pib_config_describe(b'poisson', sys.argv[2].encode(), buf, 2048)\n"
            "    print('RC', rc, lib.pib_last_error().decode() if rc else buf.value.decode())\n")
    out = subprocess.run([sys.executable, "-c", code, hook_so, AMGX_POISSON], capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RC ")]
    assert len(lines) == 3, out.stdout + out.stderr
    assert lines[0].startswith("RC 76 ") and "C++ exception: drill" in lines[0]   # PETSC_ERR_LIB
    assert lines[1].startswith("RC 55 ")                                          # PETSC_ERR_MEM
    assert lines[2].startswith("RC 0 ") and "method=cg" in lines[2]
    # the product library: no hook
    out = subprocess.run([sys.executable, "-c", code, B.library_path(), AMGX_POISSON], capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RC ")]
    assert len(lines) == 3 and all(l.startswith("RC 0 ") for l in lines), out.stdout + out.stderr


def test_config_describe_reports_the_steps_the_cycle_runs(capi):
    """`pib_sweep_pairs` (default 1) reads a sweep of the solver file as a fused pair of damped-Jacobi steps: the description
    says so next to the file's own counts; Chebyshev smoothing counts its polynomial degree instead."""
    d = capi.config_describe("poisson", AMGX_POISSON)
    assert (d["presweeps"], d["postsweeps"], d["sweep_pairs"]) == ("1", "1", "1")
    assert (d["effective_presteps"], d["effective_poststeps"]) == ("2", "2")
    d = capi.config_describe("poisson", AMGX_POISSON + "pib_sweep_pairs=0\n")
    assert (d["effective_presteps"], d["effective_poststeps"]) == ("1", "1")


def test_norm_and_chebyshev_degree_keys(capi):
    """`pib_norm` overrides the flavour's monitored norm (AmgX-style files: the L2 norm of the true residual, `norm=L2`; PETSc-options
    files: KSPCG's preconditioned norm); `pib_cheby_degree` is the degree of one Chebyshev smoothing "sweep" in the PETSc-options
    flavour, which has no native key for it (AmgX-style files say `chebyshev_polynomial_order`)."""
    d = capi.config_describe("poisson", AMGX_POISSON)
    assert d["norm"] == "unpreconditioned"
    assert capi.config_describe("poisson", AMGX_POISSON + "pib_norm=PRECONDITIONED\n")["norm"] == "preconditioned"
    p = capi.config_describe("poisson", PETSC_BOTH)
    assert p["norm"] == "preconditioned"
    assert capi.config_describe("poisson", PETSC_BOTH + "-poisson_ksp_norm_type unpreconditioned\n")["norm"] == "unpreconditioned"  # (its own spelling there)
    c = capi.config_describe("poisson", PETSC_BOTH + "-poisson_pib_smoother CHEBYSHEV\n-poisson_pib_cheby_degree 3\n")
    assert (c["smoother"], c["cheby_degree"], c["effective_presteps"]) == ("chebyshev", "3", "3")


def test_collecting_the_gpu_suite_does_not_import_torch():
    """tests/conftest.py runs the GPU suite on /opt/rocm's own HIP runtime (PIB_TORCH_FIRST=0): that only holds while no test module
    imports torch at collection -- afterwards torch would load the copies it bundles beside the ones the library already uses."""
    code = ("import sys, pytest\n"
            "class P:\n"
            "    def pytest_collection_finish(self, session):\n"
            "        print('TORCH_AT_COLLECTION', 'torch' in sys.modules, len(session.items))\n"
            "pytest.main(['tests', '-m', 'gpu', '--collect-only', '-q'], plugins=[P()])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300).stdout
    line = [l for l in out.splitlines() if l.startswith("TORCH_AT_COLLECTION")]
    assert line and line[0].split()[1] == "False" and int(line[0].split()[2]) > 600, out[-2000:]


def test_gpu_tests_never_load_torchs_hip_runtime_into_the_pytest_process():
    """tests/conftest.py: the GPU suite runs the library on /opt/rocm's own HIP runtime (PIB_TORCH_FIRST=0); a test that imports
    torch in the pytest process pulls torch's bundled runtime in beside it -- round 6 saw exactly that end in `double free or
    corruption` at interpreter exit.  GPU tests that need torch (bench.py under the launcher) start it in a child process."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    bad = []
    for f in sorted(glob.glob(os.path.join(root, "test_gpu_*.py"))):
        for k, line in enumerate(open(f), 1):
            if re.match(r"\s*(import torch|from torch)", line):
                bad.append(f"{os.path.basename(f)}:{k}")
    assert bad == []


def test_every_solver_file_key_is_exercised_by_a_test_and_there_are_at_most_35():
    """The backend's own solver-file keys (csrc/config.cpp) stay a SMALL surface: at most 35, each one used by at least one test
    of this directory (the `-<prefix>_pib_<key>` spelling of the PETSc-options flavour counts) -- a key nothing tests is a code
    path nothing tests."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = open(os.path.join(root, "petibm_amd", "csrc", "config.cpp")).read()
    keys = sorted(set(re.findall(r'get\("(pib_[a-z0-9_]+)"', cfg)) | set(re.findall(r'"default", "(pib_[a-z0-9_]+)"', cfg)))
    assert len(keys) <= 35, keys
    me = os.path.abspath(__file__)
    text = "\n".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "*.py")) if os.path.abspath(f) != me)
    text += "\n" + "\n".join(line for line in open(me) if "AMGX_" in line or "PETSC_" in line or "config_describe" in line)
    unused = [k for k in keys if not re.search(k + r"(?![a-z0-9_])", text)]
    assert unused == [], f"keys no test uses: {unused}"
