#!/usr/bin/env python3
"""bench.py -- Poisson DOF/s + CG iterations/s on the 512^3 lid-driven-cavity
pressure system (BASELINE.json metric), one rank per GPU.

A "step" is one `pSolver->solve(dP, rhs2)` (applications/navierstokes/
navierstokes.cpp:566-580) on the 512^3 DBNG operator to a relative residual of
1e-10 (north-star tolerance), zero initial guess, inputs already resident in
HBM.  N > 1: the same 512^3 problem split into DMDA-style z-slabs (strong
scaling), RCCL halo planes + scalar all-reduce.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0 (contract in the task statement) with the extra
objects `roofline` (CSR SpMV, algorithmic bytes / HIP-event launch duration)
and `cpu_baseline` (the oracle = CPU restatement of the reference path, timed
on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def solver_config(pc: str, tol: float, max_iters: int, omega: float = 0.9, pre: int = 1, post: int = 1,
                  smoother: str = "jacobi") -> str:
    prec = {"gmg": "AMG", "jacobi": "BLOCK_JACOBI", "none": "NOSOLVER"}[pc]
    if smoother == "chebyshev":
        return (f"config_version=2\nsolver(solv)=PCG\nsolv:max_iters={max_iters}\nsolv:monitor_residual=1\n"
                f"solv:convergence=RELATIVE_INI\nsolv:tolerance={tol}\nsolv:norm=L2\nsolv:store_res_history=1\n"
                f"solv:preconditioner(prec)={prec}\nprec:relaxation_factor=1.0\n"
                f"prec:cycle=V\nprec:presweeps={pre}\nprec:postsweeps={post}\nprec:smoother(smooth)=CHEBYSHEV_POLY\n"
                "smooth:chebyshev_polynomial_order=2\nsmooth:cheby_max_lambda=2.0\nsmooth:cheby_min_lambda=0.5\n"
                "pib_initial_guess_nonzero=0\n")
    return (f"config_version=2\nsolver(solv)=PCG\nsolv:max_iters={max_iters}\nsolv:monitor_residual=1\n"
            f"solv:convergence=RELATIVE_INI\nsolv:tolerance={tol}\nsolv:norm=L2\nsolv:store_res_history=1\n"
            f"solv:preconditioner(prec)={prec}\nprec:relaxation_factor=1.0\n"
            f"prec:cycle=V\nprec:presweeps={pre}\nprec:postsweeps={post}\nprec:smoother(smooth)=BLOCK_JACOBI\n"
            f"smooth:relaxation_factor={omega}\npib_initial_guess_nonzero=0\npib_sweep_pairs=0\n")


def reference_solver_file(tol: float, literal_sweeps: bool) -> str:
    """The Poisson solver file an unchanged PetIBM hands a `type: GPU` solver, key for key the one of
    examples/navierstokes/taylorgreenvortex3dRe1600_GPU/config/poisson_solver.info (and of every other *_GPU example: PCG,
    classical AMG, presweeps = postsweeps = 1, BLOCK_JACOBI 0.9, DENSE_LU coarse solver) -- with the bench's convergence
    criterion (relative 1e-10 instead of the file's absolute 1e-14) and the zero guess of the timed loop.  The AMG keys that
    describe AmgX's own coarsening (selector, interpolator, strength ...) are read and ignored by the geometric stand-in.
    literal_sweeps: pib_sweep_pairs=0, the file's sweep counts as they stand."""
    keys = [("config_version", "2"), ("communicator", "MPI_DIRECT"), ("min_rows_latency_hiding", "-1"),
            ("matrix_consolidation_lower_threshold", "0"), ("matrix_consolidation_upper_threshold", "1000"),
            ("fine_level_consolidation", "0"), ("determinism_flag", "1"), ("solver(solv)", "PCG"), ("solv:max_iters", "1000"),
            ("solv:monitor_residual", "1"), ("solv:convergence", "RELATIVE_INI"), ("solv:tolerance", f"{tol:.1E}"),
            ("solv:norm", "L2"), ("solv:print_solve_stats", "0"), ("solv:store_res_history", "1"), ("solv:print_grid_stats", "0"),
            ("solv:obtain_timings", "0"), ("solv:preconditioner(prec)", "AMG"), ("prec:algorithm", "CLASSICAL"),
            ("prec:max_iters", "1"), ("prec:cycle", "V"), ("prec:presweeps", "1"), ("prec:postsweeps", "1"),
            ("prec:max_levels", "100"), ("prec:min_coarse_rows", "2"), ("prec:interp_max_elements", "-1"),
            ("prec:interp_truncation_factor", "1.1"), ("prec:interpolator", "D2"), ("prec:max_row_sum", "1.1"),
            ("prec:selector", "PMIS"), ("prec:strength", "AHAT"), ("prec:strength_threshold", "0.25"),
            ("prec:coarse_solver(c_solver)", "DENSE_LU_SOLVER"), ("prec:dense_lu_num_rows", "128"), ("prec:dense_lu_max_rows", "0"),
            ("prec:coarsest_sweeps", "1"), ("prec:smoother(smooth)", "BLOCK_JACOBI"), ("smooth:relaxation_factor", "0.9")]
    text = "\n".join(f"{k}={v}" for k, v in keys) + "\npib_initial_guess_nonzero=0\n"
    return text + ("pib_sweep_pairs=0\n" if literal_sweeps else "")


def slab(nplanes: int, nranks: int, rank: int):
    b = 0
    for r in range(rank):
        b += nplanes // nranks + (1 if (nplanes % nranks) > r else 0)
    return b, b + nplanes // nranks + (1 if (nplanes % nranks) > rank else 0)


def manufactured_solution(n: int, k0: int, k1: int) -> np.ndarray:
    """x* = cos(pi x) cos(pi y) cos(pi z) at pressure-cell centres of the unit
    cube; zero mean by symmetry; slab [k0, k1)."""
    h = 1.0 / n
    c = np.cos(np.pi * (np.arange(n) + 0.5) * h)
    cz = c[k0:k1]
    return (cz[:, None, None] * c[None, :, None] * c[None, None, :]).reshape(-1)


def petsc_config():
    """Where a PETSc installation is, if the host has one: (cflags, ldflags, description) from `pkg-config petsc` /
    `pkg-config PETSc` or from PETSC_DIR[/PETSC_ARCH]; None when there is none (every box this has run on so far)."""
    import shutil
    import subprocess
    if shutil.which("pkg-config"):
        for name in ("petsc", "PETSc"):
            try:
                c = subprocess.run(["pkg-config", "--cflags", name], capture_output=True, text=True)
                l = subprocess.run(["pkg-config", "--libs", name], capture_output=True, text=True)
                if c.returncode == 0 and l.returncode == 0:
                    return c.stdout.split(), l.stdout.split(), f"pkg-config {name}"
            except OSError:
                pass
    d = os.environ.get("PETSC_DIR")
    if d and os.path.isdir(d):
        arch = os.environ.get("PETSC_ARCH", "")
        inc = [f"-I{os.path.join(d, 'include')}"] + ([f"-I{os.path.join(d, arch, 'include')}"] if arch else [])
        libdir = os.path.join(d, arch, "lib") if arch else os.path.join(d, "lib")
        if os.path.isdir(libdir):
            return inc, [f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-lpetsc"], f"PETSC_DIR={d} PETSC_ARCH={arch}"
    return None


def petsc_probe() -> str:
    cfg = petsc_config()
    if cfg is None:
        return "not found on this host (no pkg-config petsc, PETSC_DIR unset): port row only"
    return f"found ({cfg[2]}): the KSPSolve row (tools/petsc_ksp_driver.c) is built and run beside the port"


def petsc_baseline(n: int, tol: float, dt: float, budget_s: float = 60.0):
    """SURVEY.md 8d / BASELINE.md 4: the reference's own CPU path -- PETSc's KSPCG with -pc_type gamg
    (src/linsolver/linsolverksp.cpp:48-107; the options of examples/navierstokes/liddrivencavity2dRe100/config/
    poisson_solver.info) -- on the same int32 CSR and right-hand side the port row solved, ONLY when a PETSc installation
    is found: tools/petsc_ksp_driver.c is compiled against it at run time (mpicc, else gcc).  Returns a cpu_baseline-shaped
    dict of kind "petsc", or None when there is no PETSc (the line stays as it was)."""
    import shutil
    import subprocess
    import tempfile
    cfg = petsc_config()
    if cfg is None:
        return None
    from oracle import clib
    cc = shutil.which("mpicc") or shutil.which("gcc")
    tmp = tempfile.mkdtemp(prefix="pib_petsc_")
    exe = os.path.join(tmp, "petsc_ksp_driver")
    build = subprocess.run([cc, "-O2", os.path.join(ROOT, "tools", "petsc_ksp_driver.c"), "-o", exe] + cfg[0] + cfg[1] + ["-lm"],
                           capture_output=True, text=True)
    if build.returncode != 0:
        return {"kind": "petsc", "value": None, "note": f"PETSc found ({cfg[2]}) but the driver did not build: {build.stderr[-400:]}"}
    w = np.full(n, 1.0 / n)
    rp, cl, vl = clib.assemble_poisson32((n, n, n), [w, w, w], dt)
    xs = manufactured_solution(n, 0, n)
    b = np.empty(n ** 3)
    clib.spmv32(n ** 3, rp, cl, vl, xs, b)
    path = os.path.join(tmp, "system.bin")
    with open(path, "wb") as f:
        np.array([n ** 3, len(cl)], dtype=np.int64).tofile(f)
        np.asarray(rp, dtype=np.int32).tofile(f)
        np.asarray(cl, dtype=np.int32).tofile(f)
        np.asarray(vl, dtype=np.float64).tofile(f)
        b.tofile(f)
    opts = ["-poisson_ksp_type", "cg", "-poisson_ksp_rtol", f"{tol:g}", "-poisson_ksp_atol", "1e-50", "-poisson_ksp_max_it", "20000",
            "-poisson_pc_type", "gamg", "-poisson_pc_gamg_type", "agg", "-poisson_pc_gamg_agg_nsmooths", "1"]
    try:
        run = subprocess.run([exe, path] + opts, capture_output=True, text=True, timeout=budget_s * 4)
        line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
        r = json.loads(line)
    except Exception as e:  # noqa: BLE001
        return {"kind": "petsc", "value": None, "note": f"PETSc driver failed: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": n ** 3 / r["seconds"], "unit": "DOF/s", "cores": 1, "kind": "petsc",
            "sample": f"PETSc KSPCG + PCGAMG (agg, 1 smoothing step; the reference's poisson_solver.info) on the {n}^3 cavity Poisson "
                      f"system, rtol {tol:g}, one process: {r['iters']} iterations in {r['seconds']:.2f} s (KSPSetUp not counted), "
                      f"reason {r['reason']}, preconditioned residual {r['residual']:.3e}; {cfg[2]}",
            "iters": r["iters"], "seconds": r["seconds"], "grid": n}


def cpu_baseline(n_gpu: int, tol: float, dt: float, pre: int = 2, post: int = 2, omega: float = 0.9, budget_s: float = 45.0):
    """The oracle (CPU restatement of the same path: int32 CSR SpMV + KSPCG recurrences + the same V-cycle,
    oracle/csrc/*.c, OpenMP over every core the process may use: affinity and cgroup quota) timed on this box.  It runs the SAME workload as the GPU when
    a calibration solve at n/2 says it fits the time budget and the host has the memory; otherwise the
    bounded sample is the largest power-of-two cavity that does."""
    from oracle import clib
    clib.set_threads(clib.usable_cores())  # every core this process may use (affinity and cgroup quota respected)
    cores = clib.num_threads()

    def run(n):
        w = np.full(n, 1.0 / n)
        t0 = time.perf_counter()
        rp, cl, vl = clib.assemble_poisson32((n, n, n), [w, w, w], dt)
        g = clib.GMG((n, n, n), [w, w, w], dt, nullspace=1, pre=pre, post=post, omega=omega, coarsest_sweeps=32)
        xs = manufactured_solution(n, 0, n)
        b = np.empty(n ** 3)
        clib.spmv32(n ** 3, rp, cl, vl, xs, b)
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        r = clib.pcg_gmg32(g, rp, cl, vl, b, rtol=tol, maxit=200)
        return time.perf_counter() - t0, r, t_setup

    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    n = min(128, n_gpu)
    t, r, ts = run(n)
    while 2 * n <= n_gpu:
        need = 40.0 * (2 * n) ** 3 * 8  # ~40 doubles per cell: CSR + vectors + levels
        if 8.5 * (t + ts) > budget_s or need > 0.6 * avail:
            break
        n *= 2
        t, r, ts = run(n)
    same = "the same workload" if n == n_gpu else f"a bounded sample of it ({n}^3 instead of {n_gpu}^3)"
    return {"value": n ** 3 / t, "unit": "DOF/s", "cores": cores, "kind": "port", "petsc": petsc_probe(),
            "sample": f"{same}: GMG-PCG (oracle/csrc: KSPCG recurrences + the build's V({pre},{post}) cycle, int32 CSR) on the "
                      f"{n}^3 cavity Poisson system, rtol {tol:g}: {r['iters']} iterations in {t:.2f} s "
                      f"(+ {ts:.1f} s CPU assembly, not counted)",
            "iters": r["iters"], "seconds": t, "grid": n}


def measured_traffic(n: int, world: int):
    """HBM bytes per SpMV launch from the COMMITTED rocprofv3 PMC passes (profiles/spmv_pmc.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc passes of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950) -- a number read from a file, not measured in this run: the second value says so.
    (None, None) when no pass matches this configuration."""
    path = os.path.join(ROOT, "profiles", "spmv_pmc.json")
    try:
        for e in json.load(open(path)):
            if e.get("n") == n and e.get("gpus") == world:
                return e.get("traffic_bytes"), (f"profiles/spmv_pmc.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round "
                                                f"{e.get('round')}, {e.get('kernel')}; not collected in this run)")
    except Exception:
        pass
    return None, None


def spmv_only(args) -> int:
    """child of collect_traffic(): assemble the system and launch the CSR SpMV a few times -- what rocprofv3 --pmc counts"""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    n = args.n
    s = LinSolverHIP("poisson", config_text=solver_config(args.pc, args.tol, args.max_iters, args.omega, args.presweeps, args.postsweeps))
    w = np.full(n, 1.0 / n)
    s.assemblePoisson((n, n, n), [w, w, w], 5e-4 if n == 512 else 1e-3, capi.NULLSPACE_CONSTANT)
    s.timeKernel(0, 3)
    s.destroy()
    # ... and the plain int32-column CSR kernel (SURVEY 8d's 104 B/row, what north_star quotes): k_spmv_lds<...>
    s = LinSolverHIP("poisson", config_text=solver_config(args.pc, args.tol, args.max_iters, args.omega, args.presweeps, args.postsweeps) +
                     "pib_compress_columns=0\n")
    s.assemblePoisson((n, n, n), [w, w, w], 5e-4 if n == 512 else 1e-3, capi.NULLSPACE_CONSTANT)
    s.timeKernel(0, 3)
    s.destroy()
    return 0


def collect_traffic(n: int, mode: str):
    """`roofline.traffic` measured in THIS run: the SpMV leg once more under `rocprofv3 --pmc FETCH_SIZE` and once under
    `--pmc WRITE_SIZE` (separate passes with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; run from
    /tmp with TMPDIR=/tmp), in a child process of this script (`--spmv-only`: the product kernel in use AND the plain
    int32-column CSR kernel).  KiB per dispatch, averaged per kernel; FETCH_SIZE doubled (gfx950 tallies 128-byte requests as
    64).  ({"default": bytes, "csr_plain": bytes}, source) or (None, reason).
    mode: "auto" (when rocprofv3 is on PATH and this is not already a profiled child), "on", "off"."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if mode == "off" or os.environ.get("PIB_BENCH_CHILD") == "1":
        return None, "switched off"
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None, "rocprofv3 not found"
    env = dict(os.environ, TMPDIR="/tmp", PIB_BENCH_CHILD="1")
    kib = {"default": {}, "csr_plain": {}}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"pib_pmc_{counter}_", dir="/tmp")
        try:
            cmd = [tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--spmv-only", "--grid", str(n)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=360)
            vals = {"default": [], "csr_plain": []}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "k_spmv_lds" in name and row.get("Counter_Name") == counter:
                        plain = "k_spmv_lds_pattern" not in name and "k_spmv_lds_coded" not in name
                        vals["csr_plain" if plain else "default"].append(float(row["Counter_Value"]))
            if not vals["default"] and vals["csr_plain"]:  # (the default form IS the plain kernel: pib_compress_columns=0 in the config)
                vals["default"] = vals["csr_plain"]
            if not vals["default"]:
                return None, f"rocprofv3 --pmc {counter}: no k_spmv_lds dispatch in the output (rc {r.returncode})"
            for k in vals:
                if vals[k]:
                    kib[k][counter] = sum(vals[k]) / len(vals[k])
        except Exception as e:  # noqa: BLE001
            return None, f"rocprofv3 --pmc {counter} failed: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = {k: (2.0 * 1024.0 * v["FETCH_SIZE"] + 1024.0 * v["WRITE_SIZE"]) if len(v) == 2 else None for k, v in kib.items()}
    src = (f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes over `bench.py --spmv-only "
           f"--grid {n}`), mean per dispatch: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE; KiB per dispatch: " +
           "; ".join(f"{k}: FETCH_SIZE {v.get('FETCH_SIZE', float('nan')):.4g}, WRITE_SIZE {v.get('WRITE_SIZE', float('nan')):.4g}" for k, v in kib.items()))
    return traffic, src


def velocity_case(n: int, steps: int, warmup: int, kernel_reps: int, extra: str = "", solver: str = "PBICGSTAB", tol: float = 1e-10) -> dict:
    """The velocity solve of the same cavity, A = I/dt - c nu L (navierstokes.cpp:342-344) with PBICGSTAB + BLOCK_JACOBI to
    an absolute residual of 1e-10 (examples/navierstokes/taylorgreenvortex3dRe1600_GPU/config/velocity_solver.info),
    single GPU.  The Krylov products run matrix-free from the mesh tables (velstencil.hip: 56 B/row -- x read once, y
    written once, the quotient tables are 1-D -- instead of the CSR's 104); the roofline entry is THAT kernel group, the
    CSR SpMV of the same operator is reported next to it.  At 512^3 the CSR has 2.8e9 non-zeros (64-bit row offsets).
    solver="CHEBYSHEV": the same system and tolerance through the Chebyshev iteration (`-velocity_ksp_type chebyshev` in PETSc's
    terms, krylov.hip:solve_chebyshev: no inner products, Gershgorin bounds from the matrix) -- a solver-file choice, not what
    the reference's example files say.  It tests the TRUE residual b - A x of every iterate, whose rounding floor at 256^3 is
    ~1e-9 (eps |b_i| over 5e7 entries): its line runs to 1e-8, where BiCGStab's recurrence residual and true residual still
    agree (at the 1e-10 of the first line they no longer do: true_abs_residual 1.3e-9)."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    dt, nu = (5e-4, 1e-3) if n == 512 else (1e-3, 1e-3)
    cfg = (f"config_version=2\nsolver(solv)={solver}\nsolv:max_iters=1000\nsolv:monitor_residual=1\n"
           f"solv:convergence=ABSOLUTE\nsolv:tolerance={tol!r}\nsolv:norm=L2\nsolv:store_res_history=1\n"
           "solv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=0.9\npib_initial_guess_nonzero=0\n")
    s = LinSolverHIP("velocity", config_text=cfg + extra.replace("\\n", "\n") + "\n")
    w = np.full(n, 1.0 / n)
    a0 = np.array([[0.0 if (loc // 2) == f else -1.0 for loc in range(6)] for f in range(3)])  # all-Dirichlet cavity
    t0 = time.perf_counter()
    s.assembleVelocity((n, n, n), [w, w, w], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), a0, dt, 0.5 * nu)
    s.synchronize()
    t_setup = time.perf_counter() - t0
    UN = s.n_local
    us_d, b_d, x_d, r_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
    rng = np.random.default_rng(20260928)
    chunk = 1 << 24
    for off in range(0, UN, chunk):  # u* uniform in [-1,1), uploaded in pieces
        m = min(chunk, UN - off)
        a = rng.uniform(-1.0, 1.0, m)
        capi.check(capi.load().pib_memcpy_h2d(s._h, us_d.ptr + 8 * off, a.ctypes.data, 8 * m))
    s.matMult(us_d, b_d)
    for _ in range(warmup):
        s.solve(x_d, b_d)
    s.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(steps):
        s.solve(x_d, b_d)
        its += s.getIters()
    s.synchronize()
    el = time.perf_counter() - t0
    # residual contract with the CSR operator: |b - A x|_2 <= 1e-10 (absolute, as configured)
    s.matMult(x_d, r_d)
    res = 0.0
    for off in range(0, UN, chunk):
        m = min(chunk, UN - off)
        rb, bb = np.empty(m), np.empty(m)
        capi.check(capi.load().pib_memcpy_d2h(s._h, rb.ctypes.data, r_d.ptr + 8 * off, 8 * m))
        capi.check(capi.load().pib_memcpy_d2h(s._h, bb.ctypes.data, b_d.ptr + 8 * off, 8 * m))
        res += float(np.sum((bb - rb) ** 2))
    ms_free = s.timeKernel(5, kernel_reps)
    ms_spmv = s.timeKernel(0, max(2, kernel_reps // 4))
    rp_bytes = 8 if s.nnz >= 2 ** 31 - 1 else 4
    idx = s.productIndexBytes()
    alg_csr = spmv_algorithmic_bytes(s.nnz, UN, idx, rp_bytes)
    alg_free = 16.0 * UN  # x read once, y written once; tables are 1-D (reported apart from the CSR figure)
    # velstencil.hip vel_stencil_apply: components of >= 4 M points with lines of >= 127 take the one-launch marching form
    vel_kernel = ("pib::k_vel_product<0> (LDS-tiled march of the three components + their boundary shells in one launch: the product BiCGStab runs)"
                  if n ** 3 >= (1 << 22) and n >= 128 and "pib_fuse_velocity_product=0" not in extra and "pib_march=0" not in extra
                  else "pib::k_vel_interior4<3> / k_vel_march + k_vel_shell x 3 components (the products BiCGStab runs)")
    out = {
        "metric": f"velocity-system DOF/s ({'BiCGStab' if solver == 'PBICGSTAB' else 'Chebyshev'}+Jacobi to |r| <= {tol:g})", "value": UN * steps / el,
        "unit": "DOF/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps, "dtype": "f64",
        "config": {"workload": f"{n}^3 cavity velocity system A = I/dt - c nu L, {UN} rows, {s.nnz} nnz, "
                               f"{8 * rp_bytes}-bit row offsets, random u*"},
        "iters_per_solve": its / steps, "final_residual": s.getResidual(), "true_abs_residual": float(np.sqrt(res)),
        "setup_s": t_setup,
        "roofline": {"bound": "hbm", "kernel": vel_kernel,
                     "achieved": alg_free / ms_free / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_free / ms_free / 1e6 / HBM_PEAK_GBS, "ms_per_launch": ms_free, "algorithmic_bytes": alg_free,
                     "traffic": None},
        "csr_spmv": {"kernel": f"pib::k_spmv_lds{'_pattern' if idx == 0 else '_coded' if idx == 1 else ''}<int{8 * rp_bytes}>", "index_bytes_per_entry": idx, "achieved": alg_csr / ms_spmv / 1e6, "unit": "GB/s",
                     "frac": alg_csr / ms_spmv / 1e6 / HBM_PEAK_GBS, "ms_per_launch": ms_spmv, "algorithmic_bytes": alg_csr}}
    s.destroy()
    return out


def velocity_chebyshev_pair(n: int, kernel_reps: int) -> dict:
    """the Chebyshev line, with BiCGStab timed to the same (reachable) tolerance next to it"""
    out = velocity_case(n, 3, 1, kernel_reps, solver="CHEBYSHEV", tol=1e-8)
    ref = velocity_case(n, 3, 1, 2, solver="PBICGSTAB", tol=1e-8)
    out["bicgstab_same_tolerance"] = {k: ref[k] for k in ("ms_per_step", "iters_per_solve", "true_abs_residual")}
    return out


def velocity_bench(args):
    import torch
    assert torch.cuda.is_available()
    out = velocity_case(args.n, args.steps, args.warmup, args.kernel_reps, args.extra_config, args.velocity_solver, args.velocity_tol)
    out.update({"n_gpus": 1, "higher_is_better": True, "data": "synthetic"})
    print(json.dumps(out), flush=True)


def random_rhs_solution(n: int, k0: int, k1: int) -> np.ndarray:
    """x* uniform in [-1, 1), PCG64 seed 20260928, projected to zero mean; slab [k0, k1)"""
    rng = np.random.default_rng(20260928)
    x = rng.uniform(-1.0, 1.0, n ** 3)
    x -= x.mean()
    return x[k0 * n * n:k1 * n * n]


# Algorithmic HBM bytes per row of one multigrid-PCG iteration as this build runs it, fp64 / int32:
#   CSR SpMV 104 (12 nnz/n + 4 + 16) ; x += a p (owed by the previous iteration) and p = z + beta p in one pass 40 ;
#   r -= a w + sums: inside the V-cycle's first march, which reads w beside r and writes the new r beside x: 16 (24 as a pass of
#   its own before the end of round 2) ;
#   V(2,2) cycle on the fine level: two pre-smoothing steps from zero 16 (b read, x written), residual + restriction in one
#   march 16 + 1 (b and x read, the coarse right-hand side written; 24 + 8 + 1 as two kernels, until late in round 3),
#   prolongation + both post-smoothing steps (+ the Krylov sums) in one march 24 + 1 on the levels the marching kernels serve
#   (24 + 1 and 24 as two kernels below them, and everywhere until late in round 3) -- 16 + 17 + 25 = 58 on the finest level
#   (98), the coarser levels a seventh of that.       Sum: 227 B per row and iteration at 512^3 (272 with both detours through HBM).
def spmv_algorithmic_bytes(nnz: float, n: float, idx_bytes: int = 4, rp_bytes: float = 4.0) -> float:
    """compulsory bytes of one CSR product in the format the kernel streams: value + index per entry, row offsets, x once, y,
    and with one-byte column codes the 16-entry dictionary of every 256-row block (kernels_spmv.hip)"""
    blocks = (n + 255) // 256
    if idx_bytes == 0:  # row patterns: one byte per row; the table's index and two row offsets per 256-row block (the tables are shared)
        return 8.0 * nnz + 17.0 * n + (4.0 + 2.0 * rp_bytes) * blocks
    return (8.0 + idx_bytes) * nnz + rp_bytes * (n + 1) + 16.0 * n + (64.0 * blocks if idx_bytes == 1 else 0.0)


def solve_bytes_per_row_iter(pre: int, post: int, nnz_per_row: float, fused_residual_restrict: bool = True,
                             fused_post_pair: bool = True, n_rows: float = 134217728.0, idx_bytes: int = 4) -> float:
    spmv = (8.0 * nnz_per_row + 17.0 + 12.0 / 256.0) if idx_bytes == 0 else ((8.0 + idx_bytes) * nnz_per_row + 4.0 + 16.0 + (0.25 if idx_bytes == 1 else 0.0))
    down = (16.0 if pre >= 2 else 8.0 + 8.0) + 24.0 * max(pre - 2, 0) + (17.0 if fused_residual_restrict and pre >= 2 else 24.0 + 9.0)
    up = (25.0 if post >= 1 else 17.0) + 24.0 * max(post - 1, 0)
    # the way up, level by level (cells / 8 each): fused on the levels the marching kernels serve (pib_march_min_cells)
    up_all, cells, scale = 0.0, float(n_rows), 1.0
    while cells >= 1.0:
        up_all += (25.0 if (fused_post_pair and post == 2 and cells >= 12582912.0) else up) * scale
        cells /= 8.0
        scale /= 8.0
    return spmv + 40.0 + 16.0 + down * 8.0 / 7.0 + up_all


def poisson_case(n: int, dt: float, cfg_text: str, rhs: str, steps: int, warmup: int, kernel_reps: int, which_kernel: int = 0):
    """One single-GPU Poisson line: assemble, b = A x*, `steps` timed solves, residual contract with the CSR operator."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    s = LinSolverHIP("poisson", config_text=cfg_text)
    w = np.full(n, 1.0 / n)
    s.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_CONSTANT)
    xs_d, b_d, x_d, r_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
    xs_d.upload(manufactured_solution(n, 0, n) if rhs == "cosine" else random_rhs_solution(n, 0, n))
    s.matMult(xs_d, b_d)
    for _ in range(warmup):
        s.solve(x_d, b_d)
    s.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(steps):
        s.solve(x_d, b_d)
        its += s.getIters()
    s.synchronize()
    el = time.perf_counter() - t0
    s.matMult(x_d, r_d)
    bl = b_d.download()
    rl = bl - r_d.download()
    rel = float(np.sqrt((rl @ rl) / (bl @ bl)))
    ms_k = s.timeKernel(which_kernel, kernel_reps)
    out = {"value": n ** 3 * steps / el, "unit": "DOF/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
           "iters_per_solve": its / steps, "true_rel_residual": rel, "grid": [n, n, n], "dt": dt, "rhs": rhs}
    nnz, nl = s.nnz, s.n_local
    out["index_bytes_per_entry"] = s.productIndexBytes()
    s.destroy()
    return out, ms_k, nnz, nl


def secondary_poisson(n, dt, cfg_text, rhs, which_kernel, args) -> dict:
    out, ms_k, nnz, nl = poisson_case(n, dt, cfg_text, rhs, 2, 1, args.kernel_reps, which_kernel)
    if which_kernel == 0:
        idx = out["index_bytes_per_entry"]
        alg = spmv_algorithmic_bytes(nnz, nl, idx)
        kname = ("pib::k_spmv_lds_pattern<int32> (CSR SpMV from one-byte row patterns)" if idx == 0 else
                 "pib::k_spmv_lds_coded<int32> (CSR SpMV from one-byte column codes)" if idx == 1 else "pib::k_spmv_lds<int32> (CSR SpMV)")
    else:
        alg = 16.0 * nl + 8.0 * 3 * n  # SURVEY.md 8d B_stencil: x read once, y written once, the 1-D width arrays
        kname = "pib::k_level<0,4> (matrix-free stencil twin, reported apart from the CSR figure)"
    out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": alg / ms_k / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / ms_k / 1e6 / HBM_PEAK_GBS, "ms_per_launch": ms_k, "algorithmic_bytes": alg, "traffic": None}
    out["metric"] = "Poisson DOF/s (one pressure solve to rel. residual 1e-10)"
    return out


def host_buffer_case(n: int, dt: float, cfg_text: str) -> dict:
    """The same solve with x and b handed over as HOST arrays, the way linsolverksp.cpp:97-116 / AmgXSolver::solve
    (linsolveramgx.cpp:96-105) receive them from an application whose Vecs live in host memory: pib_solve stages b and
    the initial guess to HBM and x back over PCIe inside the timed region.  Reported apart; never the headline value."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    s = LinSolverHIP("poisson", config_text=cfg_text)
    w = np.full(n, 1.0 / n)
    s.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_CONSTANT)
    xs = manufactured_solution(n, 0, n)
    b = np.empty_like(xs)
    s.matMult(xs, b)
    x = np.zeros_like(xs)
    s.solve(x, b)  # warm-up: the staging buffers are allocated here
    steps, its, el = 2, 0, 0.0
    for _ in range(steps):
        x[:] = 0.0
        t0 = time.perf_counter()
        s.solve(x, b)  # returns with x back in host memory
        el += time.perf_counter() - t0
        its += s.getIters()
    r = np.empty_like(xs)
    s.matMult(x, r)
    r = b - r
    rel = float(np.sqrt((r @ r) / (b @ b)))
    s.destroy()
    return {"metric": "Poisson DOF/s, x and b in pageable host memory (PCIe inclusive)", "value": n ** 3 * steps / el, "unit": "DOF/s",
            "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": 1, "iters_per_solve": its / steps, "true_rel_residual": rel,
            "grid": [n, n, n], "dt": dt, "rhs": "cosine", "staged": "b in, x out (8 n^3 bytes each); x in as well when the solver file keeps the guess"}


def pinned_device_case(n: int, dt: float, cfg_text: str, steps: int = 2) -> dict:
    """The pinned-row convention (MatZeroRowsColumns on row 0: every `type: GPU` run of PetIBM, navierstokes.cpp:414-420) on
    the device-assembled operator with x and b resident in HBM: the headline's solve under the other null-space convention."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    s = LinSolverHIP("poisson", config_text=cfg_text)
    w = np.full(n, 1.0 / n)
    s.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_PINNED)
    xs = manufactured_solution(n, 0, n)
    xs -= xs[0]
    xs_d, b_d, x_d, r_d = s.deviceVec(), s.deviceVec(), s.deviceVec(), s.deviceVec()
    xs_d.upload(xs)
    s.matMult(xs_d, b_d)
    s.solve(x_d, b_d)
    s.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(steps):
        s.solve(x_d, b_d)
        its += s.getIters()
    s.synchronize()
    el = time.perf_counter() - t0
    s.matMult(x_d, r_d)
    bl = b_d.download()
    rl = bl - r_d.download()
    rel = float(np.sqrt((rl @ rl) / (bl @ bl)))
    fused = int(s.counters()[6])
    s.destroy()
    return {"metric": "Poisson DOF/s, pinned pressure row (the reference's type: GPU convention), x and b in HBM", "value": n ** 3 * steps / el,
            "unit": "DOF/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": 1, "iters_per_solve": its / steps,
            "true_rel_residual": rel, "grid": [n, n, n], "dt": dt, "rhs": "cosine (x*[0] = 0)", "residual_updates_in_vcycle": fused}


def dropin_amgx_route_case(n: int, dt: float, tol: float, register_host: bool = False, steps: int = 2) -> dict:
    """What an UNCHANGED PetIBM does with a `type: GPU` Poisson solver, end to end (src/linsolver/linsolveramgx.cpp:84,96 behind
    include/petibm_amd/AmgXSolver.hpp): a host int32 CSR whose row / column 0 MatZeroRowsColumns has replaced by the identity
    (navierstokes.cpp:414-420; the pattern keeps its explicit zeros) goes through pib_set_csr_i32 ONLY -- no grid hint, no device
    assembly: the backend recovers the mesh structure from the entries --, the solver file is the reference's own
    (examples/navierstokes/taylorgreenvortex3dRe1600_GPU/config/poisson_solver.info, with the bench's tolerance), b[0] = 0
    (navierstokes.cpp:553-558), x and b are pageable host arrays and x is the initial guess (AmgXSolver::solve).  The host CSR is
    obtained by downloading a device-assembled copy (the bench's way of having an application's matrix at this size within
    seconds); that solver is destroyed before the timed one is created.
    register_host: the caller's x / b page-locked with hipHostRegister around the solves (what an application could do once)."""
    import ctypes
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    w = np.full(n, 1.0 / n)
    t = LinSolverHIP("scratch", config_text=solver_config("jacobi", tol, 10))
    t.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_PINNED)
    rp, cl, vl = t.getCSR()
    rp32, cl32 = rp.astype(np.int32), cl.astype(np.int32)
    del rp, cl
    xs = manufactured_solution(n, 0, n)
    xs -= xs[0]
    b = np.empty_like(xs)
    t.matMult(xs, b)
    t.destroy()
    assert b[0] == 0.0 and vl[0] == 1.0
    text = reference_solver_file(tol, False).replace("pib_initial_guess_nonzero=0\n", "")  # x is the guess, as AmgX takes it
    s = LinSolverHIP("poisson", config_text=text)

    class Host:  # (setMatrix wants .rowptr / .col / .val)
        rowptr, col, val = rp32, cl32, vl
    t0 = time.perf_counter()
    s.setMatrix(Host)
    s.synchronize()
    t_set = time.perf_counter() - t0
    st = s.gridStructure()
    x = np.zeros_like(xs)
    hip = None
    if register_host:
        hip = ctypes.CDLL("libamdhip64.so")
        for a in (x, b):
            rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 0)
            if rc != 0:
                raise RuntimeError(f"hipHostRegister failed ({rc})")
    s.solve(x, b)  # warm-up: the staging buffers are allocated here
    its, el, h2d, d2h = 0, 0.0, 0.0, 0.0
    for _ in range(steps):
        x[:] = 0.0
        t0 = time.perf_counter()
        s.solve(x, b)  # returns with x back in host memory
        el += time.perf_counter() - t0
        its += s.getIters()
        sm = s.stagingMs()
        h2d, d2h = h2d + sm[0], d2h + sm[1]
    fused = int(s.counters()[6])
    r = np.empty_like(xs)
    s.matMult(x, r)
    r = b - r
    rel = float(np.sqrt((r @ r) / (b @ b)))
    err = float(np.abs(x - xs).max() / np.abs(xs).max())
    if hip is not None:
        for a in (x, b):
            hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data))
    s.destroy()
    return {"metric": "Poisson DOF/s through the reference's AmgX plug point: host int32 CSR with pinned row 0 via pib_set_csr_i32 only, "
                      "the reference's solver file, pageable host x / b, x as the guess (PCIe inclusive)",
            "value": n ** 3 * steps / el, "unit": "DOF/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": 1,
            "iters_per_solve": its / steps, "true_rel_residual": rel, "max_rel_error_vs_manufactured": err, "set_matrix_s": t_set,
            "structure_recovered": bool(st and st.get("detected")), "nullspace_detected": (st or {}).get("nullspace"),
            "residual_updates_in_vcycle": fused, "host_buffers": "hipHostRegister'ed" if register_host else "pageable",
            "copy_in_ms": h2d / steps, "copy_out_ms": d2h / steps, "device_ms": 1e3 * el / steps - (h2d + d2h) / steps,
            "staged": "b and the guess in, x out (8 n^3 bytes each)", "grid": [n, n, n], "dt": dt, "rhs": "cosine (x*[0] = 0)",
            "cycle": "V(1,1) of the file read as fused pairs: V(2,2) (pib_sweep_pairs=1, default)"}


def refuse(args, why: str) -> int:
    """A run that cannot start still prints ONE JSON line (value null + the reason) instead of a bare traceback."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": f"Poisson DOF/s (one pressure solve to rel. residual 1e-10), {args.n}^3 cavity",
                          "value": None, "unit": "DOF/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "f64", "data": "synthetic", "config": {"workload": f"{args.n}^3 cavity pressure Poisson"},
                          "notes": [why]}), flush=True)
    return 2


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one per GPU, RCCL over xGMI) through
    `python -m torch.distributed.run` on 127.0.0.1 with a free port, and pass rank 0's JSON line through."""
    import socket
    import subprocess
    try:
        import torch
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:  # noqa: BLE001
        ndev = 0
    if ndev < args.gpus and os.environ.get("PIB_BENCH_SHARE_GPU", "0") != "1":
        return refuse(args, f"--gpus {args.gpus} but only {ndev} GPU(s) visible on this node (one rank per GPU)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", dest="n", type=int, default=512, help="cells per direction (BASELINE: 512)")
    ap.add_argument("--pc", default="gmg", choices=["gmg", "jacobi", "none"])
    ap.add_argument("--tol", type=float, default=1e-10)
    ap.add_argument("--max-iters", type=int, default=20000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (random RHS, 256^3, stencil twin, velocity)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--omega", type=float, default=0.9, help="Jacobi smoother relaxation factor of the V-cycle")
    ap.add_argument("--smoother", default="jacobi", choices=["jacobi", "chebyshev"])
    ap.add_argument("--extra-config", default="", help="extra solver-config lines (\\n separated), e.g. pib_coarse_tail=0")
    # V(2,2) with damped Jacobi: 11 PCG iterations and 119 ms per 512^3 solve; V(1,1): 15 and 130 ms, V(3,3): 10 and 127 ms
    ap.add_argument("--presweeps", type=int, default=2)
    ap.add_argument("--postsweeps", type=int, default=2)
    ap.add_argument("--transport", default=os.environ.get("PIB_TRANSPORT", "auto"), choices=["auto", "rccl", "peer"],
                    help="N > 1: auto (default: RCCL and the peer windows are both timed on untimed solves, the faster one that meets "
                         "the residual contract runs), rccl (falls back to peer if it fails), or peer (HIP-IPC-mapped neighbours, one "
                         "node; also what lets several ranks share one GPU with PIB_BENCH_SHARE_GPU=1)")
    ap.add_argument("--velocity-solver", default="PBICGSTAB", choices=["PBICGSTAB", "CHEBYSHEV"],
                    help="--system velocity: the Krylov method of the solver file (the reference's examples: PBICGSTAB)")
    ap.add_argument("--velocity-tol", type=float, default=1e-10, help="--system velocity: absolute tolerance of the solve")
    ap.add_argument("--system", default="poisson", choices=["poisson", "velocity"],
                    help="poisson (the BASELINE metric) or the velocity system A = I/dt - c nu L with BiCGStab+Jacobi")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from rocprofv3 --pmc passes of the SpMV leg made in THIS run (N = 1; auto: when rocprofv3 is on "
                         "PATH); otherwise the committed passes of profiles/spmv_pmc.json, labelled as such")
    ap.add_argument("--no-tune-recurrence", dest="tune_recurrence", action="store_false",
                    help="N > 1: do not time the single-reduction CG recurrence against the standard one on untimed solves")
    ap.add_argument("--spmv-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.spmv_only:
        return spmv_only(args)
    if args.system == "velocity":
        return velocity_bench(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    try:
        return poisson_bench(args)
    except Exception as exc:  # noqa: BLE001 -- e.g. the communicator cannot be set up: still ONE JSON line with the reason
        import traceback
        traceback.print_exc()
        refuse(args, f"{type(exc).__name__}: {exc}")
        return 1


def poisson_bench(args) -> int:
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        return refuse(args, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        return refuse(args, "no GPU visible: bench.py has no CPU path")
    share_probe = os.environ.get("PIB_BENCH_SHARE_GPU", "0") == "1"
    if not share_probe and torch.cuda.device_count() < (world if world > 1 else 1):
        return refuse(args, f"{world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU)")
    # PIB_BENCH_SHARE_GPU=1 (testing only): all ranks on cuda:0, torch side on gloo -- the N > 1 code path on a 1-GPU
    # box; RCCL refuses several ranks per device, the peer transport (--transport peer) takes them.
    share = os.environ.get("PIB_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    notes = []
    # ---- the torch side of an N > 1 run (timing barrier, max over ranks, the id's broadcast): nccl (= RCCL) where it comes
    # up -- proved by a first collective, the bootstrap is lazy --, gloo otherwise.  First contact with a multi-GPU node must
    # not end in a traceback: every fallback taken is named in the line's `notes`.
    cpu_side = share
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            try:
                if os.environ.get("PIB_FORCE_RCCL_FAIL") == "1":
                    raise RuntimeError("forced failure (PIB_FORCE_RCCL_FAIL=1)")
                import datetime
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all-reduce of ones over {world} ranks gave {probe.item()}")
            except Exception as e:  # noqa: BLE001
                notes.append(f"torch.distributed backend nccl (RCCL) failed on first contact ({type(e).__name__}: {e}): gloo carries the "
                             "bench's own barrier / max-over-ranks instead")
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                dist.init_process_group(backend="gloo")
                cpu_side = True
    red_dev = "cpu" if (world > 1 and cpu_side) else "cuda"
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(ok: bool) -> bool:
        """every rank's verdict: a rank that failed must not leave the others inside a collective"""
        if world == 1:
            return ok
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=red_dev)
        dist.all_reduce(t)
        return int(t.item()) == 0

    def make_id(transport: str):
        import ctypes
        buf = ctypes.create_string_buffer(capi.UID_BYTES)
        if rank == 0:
            capi.check((capi.load().pib_comm_peer_id if transport == "peer" else capi.load().pib_comm_unique_id)(buf))
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
        if not cpu_side:
            t = t.cuda()
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    n = args.n
    dt = 5e-4 if n == 512 else 1e-3  # cfg3 (512^3) dt=5e-4, cfg2 (256^3) dt=1e-3
    pN = n ** 3
    k0, k1 = slab(n, world, rank)
    base_text = solver_config(args.pc, args.tol, args.max_iters, args.omega, args.presweeps, args.postsweeps,
                              args.smoother) + args.extra_config.replace("\\n", "\n") + "\n"

    def setup(transport: str, extra: str = ""):
        """solver, system, vectors and the warm-up solves on one transport; (state, error)"""
        st = {}
        try:
            uid = make_id(transport) if world > 1 else None
            if transport == "rccl" and os.environ.get("PIB_FORCE_RCCL_FAIL") == "1":
                # first-contact drill (tests, tools/first_contact.sh): behave as if RCCL's bootstrap had failed on this node -- raised
                # HERE, where pib_create's error would surface; the library itself carries no test hook
                raise RuntimeError("ncclCommInitRank: forced failure (PIB_FORCE_RCCL_FAIL=1)")
            sv = LinSolverHIP("poisson", config_text=base_text + extra, rank=rank, nranks=world, uid=uid, device=local)
            st["s"] = sv
            w = np.full(n, 1.0 / n)
            t_s = time.perf_counter()
            sv.assemblePoisson((n, n, n), [w, w, w], dt, capi.NULLSPACE_CONSTANT)
            sv.synchronize()
            st["t_setup"] = time.perf_counter() - t_s
            xs = manufactured_solution(n, k0, k1)
            st["xs_d"], st["b_d"], st["x_d"] = sv.deviceVec(), sv.deviceVec(), sv.deviceVec()
            st["xs_d"].upload(xs)
            sv.matMult(st["xs_d"], st["b_d"])  # b = DBNG x*  (compatible with the constant null space)
            del xs
            for _ in range(args.warmup):
                sv.solve(st["x_d"], st["b_d"])
            sv.synchronize()
            return st, None
        except Exception as e:  # noqa: BLE001
            return st, e

    def timed(state, reps=2):
        barrier()
        t_a = time.perf_counter()
        for _ in range(reps):
            state["s"].solve(state["x_d"], state["b_d"])
        barrier()
        tt = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) / reps

    def checked(state) -> bool:
        """a candidate's last solve meets the residual contract (recomputed with the CSR operator) and every rank agrees"""
        try:
            r_c = state["s"].deviceVec()
            state["s"].matMult(state["x_d"], r_c)
            bl_c = state["b_d"].download()
            rl_c = bl_c - r_c.download()
            loc = [float(rl_c @ rl_c), float(bl_c @ bl_c)]
        except Exception:  # noqa: BLE001
            loc = [float("inf"), 1.0]
        t = torch.tensor(loc, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)
        return bool(torch.isfinite(t[0])) and float(torch.sqrt(t[0] / t[1]).item()) <= 1.5 * args.tol

    def describe(e):
        return (type(e).__name__ + ": " + str(e)) if e is not None else "on another rank"

    transport = args.transport if world > 1 else "none"
    if world == 1:
        st, err = setup(transport)
        if err is not None:
            raise err
    elif transport in ("rccl", "peer"):
        st, err = setup(transport)
        if not agree(err is None):
            if transport == "rccl":
                # RCCL's bootstrap or its first collectives failed here or on another rank: the peer transport (HIP-IPC windows,
                # device-ordered flags; csrc/halo.hip) needs nothing of RCCL.  The failed solver is left alone (destroying a
                # communicator in an unknown state can hang).
                notes.append(f"--transport rccl failed on first contact ({describe(err)}): fell back to --transport peer")
                transport = "peer"
                st, err = setup(transport)
                if not agree(err is None):
                    raise RuntimeError(f"both transports failed; peer: {describe(err)}")
            else:
                raise err if err is not None else RuntimeError("set-up failed on another rank")
    else:
        # --transport auto (the default): both transports are tried on untimed solves -- RCCL's grouped send / recv and
        # all-reduce kernels, and the peer windows (direct stores into the neighbour's HBM over xGMI, flags in stream order:
        # csrc/halo.hip) -- and the faster one that meets the residual contract runs the timed region.  What a collective
        # costs between THESE GPUs decides, and no build box ever had two.
        st_r, err_r = setup("rccl")
        ok_r = agree(err_r is None) and checked(st_r)
        t_r = timed(st_r) if ok_r else None
        if not ok_r:
            notes.append(f"--transport rccl failed on first contact ({describe(err_r)}): fell back to --transport peer")
        os.environ["PIB_PEER_TIMEOUT_S"] = str(min(float(os.environ.get("PIB_PEER_TIMEOUT_S", "90")), 90.0 if ok_r else 600.0))
        st_p, err_p = setup("peer")
        ok_p = agree(err_p is None) and checked(st_p)
        t_p = timed(st_p) if ok_p else None
        if not ok_p and ok_r:
            notes.append(f"peer transport not usable here ({describe(err_p)}): RCCL runs the timed region")
        if not ok_r and not ok_p:
            raise RuntimeError(f"both transports failed; rccl: {describe(err_r)}; peer: {describe(err_p)}")
        if ok_r and ok_p:
            notes.append(f"transport tuned at first contact: rccl {1e3 * t_r:.2f} ms, peer {1e3 * t_p:.2f} ms per solve")
        use_peer = ok_p and (not ok_r or t_p < 0.98 * t_r)
        transport = "peer" if use_peer else "rccl"
        st = st_p if use_peer else st_r
        loser = st_r if use_peer else st_p
        if (ok_r and ok_p) and "s" in loser:
            loser["s"].destroy()
    s, xs_d, b_d, x_d, t_setup = st["s"], st["xs_d"], st["b_d"], st["x_d"], st["t_setup"]

    # ---- several ranks: the CG recurrence is chosen at first contact.  The single-reduction recurrence (pib_cg_single_reduction,
    # KSPCGUseSingleReduction) trades two all-reduces per iteration for 16 B/row of vector traffic; which wins depends on what an
    # all-reduce costs between THESE GPUs, which nobody could measure before this run: both are timed on untimed solves and the
    # faster one runs the timed region (--extra-config pib_cg_single_reduction=0/1 pins it).
    recurrence = "single-reduction" if "pib_cg_single_reduction=1" in base_text else "standard"
    if world > 1 and args.pc == "gmg" and "pib_cg_single_reduction" not in base_text and args.tune_recurrence:
        t_std = timed(st)
        st2, err2 = setup(transport, "pib_cg_single_reduction=1\n")
        if agree(err2 is None):
            t_sr = timed(st2)
            notes.append(f"CG recurrence tuned at first contact: standard {1e3 * t_std:.2f} ms, single-reduction {1e3 * t_sr:.2f} ms per solve")
            if t_sr < 0.98 * t_std:
                st["s"].destroy()
                st = st2
                s, xs_d, b_d, x_d = st["s"], st["xs_d"], st["b_d"], st["x_d"]
                recurrence = "single-reduction"
            else:
                st2["s"].destroy()
        else:
            notes.append(f"single-reduction CG could not be set up ({err2 if err2 else 'on another rank'}): standard recurrence")

    barrier()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        s.solve(x_d, b_d)
        iters += s.getIters()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # Everything below is reporting around the timed value: a failure in it must not lose the line (and must not leave
    # the other ranks waiting in a collective), so local work is guarded and the collectives are unconditional.
    # residual contract, recomputed on the device with the CSR operator
    try:
        r_d = s.deviceVec()
        s.matMult(x_d, r_d)
        bl = b_d.download()
        rl = bl - r_d.download()
        local = [float(rl @ rl), float(bl @ bl)]
    except Exception as e:  # noqa: BLE001
        local = [float("nan"), 1.0]
        notes.append(f"residual check failed: {e}")
    num = torch.tensor(local, dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(num)
    true_rel = float(torch.sqrt(num[0] / num[1]).item())

    # roofline of the dominant kernel: CSR SpMV, HIP events on the solver's stream
    nnz_l, n_l = s.nnz, s.n_local
    csr_bytes = 12.0 * nnz_l + 4.0 * (n_l + 1) + 16.0 * n_l  # SURVEY.md 8d B_spmv_csr (fp64 values, int32 columns)
    idx_bytes = s.productIndexBytes()  # 1: the product streams a one-byte column code per entry (pib_compress_columns), 4: the column
    alg_bytes = spmv_algorithmic_bytes(nnz_l, n_l, idx_bytes)
    try:
        ms_spmv = s.timeKernel(0, args.kernel_reps)
        counters = s.counters()
    except Exception as e:  # noqa: BLE001
        ms_spmv, counters = float("nan"), [0] * 8
        notes.append(f"kernel timing failed: {e}")
    achieved = alg_bytes / (ms_spmv * 1e-3) / 1e9
    all_counters = [[int(c) for c in counters]]
    if world > 1:
        gathered = [torch.zeros(8, dtype=torch.int64, device=red_dev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([int(c) for c in counters], dtype=torch.int64, device=red_dev))
        all_counters = [[int(v) for v in g.cpu().tolist()] for g in gathered]

    out = None
    if rank == 0:
        # HBM traffic of the SpMV: counted in this run where rocprofv3 exists (one GPU), else the committed passes
        traffic = (None, None)
        traffic_plain = None
        if world == 1:
            traffic = collect_traffic(n, args.pmc)
            if traffic[0] is None:
                if args.pmc == "on":
                    notes.append(f"--pmc on: {traffic[1]}")
                traffic = measured_traffic(n, world)
            else:
                traffic_plain = traffic[0].get("csr_plain")
                traffic = (traffic[0].get("default"), traffic[1])
        # the kernel north_star names, in the same process: the plain int32-column CSR product (104 B per 7-point row,
        # 13.94 GB per 512^3 launch) -- pib_compress_columns=0, same matrix, HIP events on the solver's stream
        csr_plain = None
        if world == 1:
            try:
                free_b, _ = s.deviceMemInfo()
                if idx_bytes == 4:
                    ms_plain = ms_spmv
                elif free_b < 2.0 * (12.0 * nnz_l + 64.0 * n_l):  # (a second copy of the system beside the first: 768^3 and up may not have it)
                    raise RuntimeError(f"{free_b / 1e9:.0f} GB free: no room for a second copy of the matrix")
                else:
                    s2 = LinSolverHIP("poisson", config_text=base_text + "pib_compress_columns=0\n")
                    w1 = np.full(n, 1.0 / n)
                    s2.assemblePoisson((n, n, n), [w1, w1, w1], dt, capi.NULLSPACE_CONSTANT)
                    assert s2.productIndexBytes() == 4
                    ms_plain = s2.timeKernel(0, args.kernel_reps)
                    s2.destroy()
                gbs_plain = csr_bytes / (ms_plain * 1e-3) / 1e9
                csr_plain = {"kernel": "pib::k_spmv_lds<int32> (fp64 CSR SpMV, int32 columns + int32 row offsets: pib_compress_columns=0)",
                             "ms_per_launch": ms_plain, "algorithmic_bytes": csr_bytes, "bytes_per_row": csr_bytes / n_l,
                             "achieved": gbs_plain, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_plain / HBM_PEAK_GBS,
                             "traffic": traffic_plain}
            except Exception as e:  # noqa: BLE001
                notes.append(f"plain-CSR product timing failed: {e}")
        out = {
            "metric": f"Poisson DOF/s (one pressure solve to rel. residual 1e-10), {n}^3 cavity",
            "value": pN * args.steps / elapsed, "unit": "DOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}^3 lid-driven cavity pressure Poisson (DBNG, 7-point, fp64 CSR int32), "
                                   f"PCG+{args.pc}" + (f" V({args.presweeps},{args.postsweeps})" if args.pc == "gmg" else "") +
                                   f", zero guess, rtol {args.tol:g}, manufactured cosine RHS",
                       "grid": [n, n, n], "dt": dt, "parallelism": f"zslab{world}", "pc": args.pc,
                       "transport": transport, "cg_recurrence": recurrence},
            "cg_iters_per_s": iters / elapsed, "iters_per_solve": iters / args.steps,
            "true_rel_residual": true_rel, "setup_s": t_setup,
            "spmv_gdof_per_s": n_l / (ms_spmv * 1e-3) / 1e9,
            "roofline": {"bound": "hbm",
                         "kernel": ("pib::k_spmv_lds_pattern<int32> (fp64 CSR SpMV K1 from one-byte row patterns, local slab)" if idx_bytes == 0
                                    else "pib::k_spmv_lds_coded<int32> (fp64 CSR SpMV K1 from one-byte column codes, local slab)" if idx_bytes == 1
                                    else "pib::k_spmv_lds<int32> (fp64 CSR SpMV K1, local slab)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0], "traffic_source": traffic[1],
                         "ms_per_launch": ms_spmv, "algorithmic_bytes": alg_bytes,
                         # the bytes of the format the kernel streams ((8 + index_bytes) per entry + 4 per row offset + x once + y
                         # + 64 B of dictionary per 256 rows); SURVEY 8d's plain-CSR figure beside it, against the same launch time
                         "index_bytes_per_entry": idx_bytes,
                         # (a) the plain-CSR launch north_star quotes, timed in this process, its PMC traffic from the same passes
                         "csr_plain": csr_plain},
            "counters": {"spmv": int(counters[0]), "pc_apply": int(counters[1]), "reductions": int(counters[2]),
                         "halo_exchanges": int(counters[3]), "host_polls": int(counters[4]),
                         "comm_ranks": int(counters[5]),  # ncclCommCount of the solver's communicator (1: none)
                         "residual_updates_in_vcycle": int(counters[6]),  # iterations whose r -= a w ran inside the V-cycle's first march
                         "halo_bytes_sent": int(counters[7])},  # by this rank in the last solve (exchanges + all-gathers)
        }
        try:  # what runs (pib_describe) and what the bounded placement search did on this box
            out["runs"] = s.describe().splitlines()
            out["placement"] = dict(zip(("searches", "candidates", "ms_had", "ms_kept", "held_bytes", "search_ms"), s.placementInfo()))
        except Exception as e:  # noqa: BLE001
            notes.append(f"describe failed: {e}")
        if world > 1:  # what every rank communicated in its last solve
            out["per_rank"] = [{"rank": q, "halo_exchanges": int(all_counters[q][3]), "reductions": int(all_counters[q][2]),
                                "halo_bytes_sent": int(all_counters[q][7]), "spmv": int(all_counters[q][0])} for q in range(world)]
        if args.pc == "gmg":
            # the whole solve against the roofline: algorithmic bytes of every kernel of an iteration (model in
            # solve_bytes_per_row_iter; the initial residual + V-cycle count as one more iteration) / the measured time.
            # Several ranks: all rows against the ranks' combined peak (the slabs' redundant ghost planes are not counted).
            bpr = solve_bytes_per_row_iter(args.presweeps, args.postsweeps, nnz_l / n_l,
                                           "pib_fuse_residual_restrict=0" not in args.extra_config,
                                           "pib_fuse_post_pair=0" not in args.extra_config, float(pN), idx_bytes)
            per_solve = iters / args.steps + 1.0
            gbs = bpr * pN * per_solve / (elapsed / args.steps) / 1e9
            # (b) inside `roofline`, where the driver's parsed record keeps it
            out["roofline"]["solve"] = {"bound": "hbm", "bytes_per_row_per_iteration": bpr, "iterations_counted": per_solve,
                                        "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * world),
                                        "ms_per_iteration": 1e3 * elapsed / args.steps / per_solve, "ms_per_solve": 1e3 * elapsed / args.steps}
        def finite(o):  # NaN is not JSON
            if isinstance(o, dict):
                return {k: finite(v) for k, v in o.items()}
            if isinstance(o, float) and not np.isfinite(o):
                return None
            return o
        out = finite(out)
        out["cpu_baseline"] = None
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.tol, dt, args.presweeps, args.postsweeps, args.omega)
            except Exception as e:  # noqa: BLE001
                notes.append(f"cpu baseline failed: {e}")
            try:  # the reference's own library on the same sample, when the host has one (never in the build image)
                pb = petsc_baseline(out["cpu_baseline"]["grid"] if out["cpu_baseline"] else 128, args.tol, dt)
                if pb is not None:
                    out["cpu_baseline_petsc"] = pb
            except Exception as e:  # noqa: BLE001
                notes.append(f"PETSc baseline failed: {e}")
        if notes:
            out["notes"] = notes
    s.destroy()
    if rank == 0 and world == 1 and not args.no_secondary and n == 512 and args.pc == "gmg":
        # the other lines SURVEY.md 8d asks for, each with its own residual check; failures are notes, never the headline
        base_cfg = solver_config(args.pc, args.tol, args.max_iters, args.omega, args.presweeps, args.postsweeps, args.smoother)
        out["secondary"] = []
        for name, fn in (("random_rhs_512", lambda: secondary_poisson(512, 5e-4, base_cfg, "random", 0, args)),
                         ("config2_256_cubed", lambda: secondary_poisson(256, 1e-3, base_cfg, "cosine", 0, args)),
                         ("stencil_twin_products_512", lambda: secondary_poisson(512, 5e-4, base_cfg + "pib_matrix_free_poisson=1\n",
                                                                                 "cosine", 3, args)),
                         # the solver file an UNCHANGED PetIBM brings (V(1,1) in AmgX's terms): as the backend reads it by
                         # default -- a sweep = a fused pair of damped-Jacobi steps -- and with the counts taken literally
                         ("reference_solver_file_512", lambda: dict(secondary_poisson(512, 5e-4, reference_solver_file(args.tol, False), "cosine", 0, args),
                                                                    cycle="V(1,1) of the file read as fused pairs: V(2,2) (pib_sweep_pairs=1, default)")),
                         ("reference_solver_file_512_literal", lambda: dict(secondary_poisson(512, 5e-4, reference_solver_file(args.tol, True), "cosine", 0, args),
                                                                            cycle="V(1,1) literally (pib_sweep_pairs=0)")),
                         ("velocity_256_cubed", lambda: velocity_case(256, 2, 1, args.kernel_reps)),
                         ("velocity_256_cubed_chebyshev", lambda: velocity_chebyshev_pair(256, args.kernel_reps)),
                         ("host_buffers_512", lambda: host_buffer_case(512, 5e-4, base_cfg)),
                         # the reference's `type: GPU` convention: pressure row 0 pinned -- on the device-assembled operator, and
                         # end to end the way an unchanged PetIBM hands things over (host CSR through setMatrix only, host x / b)
                         ("pinned_row_512", lambda: pinned_device_case(512, 5e-4, base_cfg)),
                         ("dropin_amgx_route_512", lambda: dropin_amgx_route_case(512, 5e-4, args.tol))):
            try:
                entry = fn()
                entry["name"] = name
                out["secondary"].append(entry)
                if name == "dropin_amgx_route_512":
                    # (c) the route an unchanged PetIBM takes, split: device / copy in / copy out per solve, and setMatrix
                    out["roofline"]["dropin"] = {k: entry[k] for k in ("ms_per_step", "device_ms", "copy_in_ms", "copy_out_ms", "set_matrix_s",
                                                                      "iters_per_solve", "true_rel_residual", "host_buffers")}
            except Exception as exc:  # noqa: BLE001
                out.setdefault("notes", []).append(f"secondary line {name} failed: {exc}")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
