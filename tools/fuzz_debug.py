"""Flake hunt: the multigrid-forms fuzz case of some seeds, every form REPS times; any run whose solution differs from the
form's first run, or from the per-phase form's, is reported.  python tools/fuzz_debug.py REPS seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_parity import gmg_cfg
from test_gpu_fuzz import _multigrid_mesh

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
POISON = os.environ.get("PIB_FUZZ_POISON", "")  # "nan" / "big": dirty the device memory the next solver's hipMallocs will get


LDS = os.environ.get("PIB_FUZZ_LDS", "")  # "nan" / "big": fill every CU's LDS before each solve (tools/lds_poison.hip)
_lds = None


def poison_lds():
    global _lds
    if not LDS:
        return
    import ctypes
    if _lds is None:
        _lds = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_poison.so"))
        _lds.lds_poison.argtypes = [ctypes.c_double]
    assert _lds.lds_poison(float("nan") if LDS == "nan" else 1.0e30) == 0


def poison():
    if not POISON:
        return
    import torch
    v = float("nan") if POISON == "nan" else 1.0e30
    ts = [torch.full((1 << 25,), v, dtype=torch.float64, device="cuda") for _ in range(8)]  # 2 GB
    torch.cuda.synchronize()
    del ts
    torch.cuda.empty_cache()


for seed in [int(a) for a in sys.argv[2:]] or list(range(16)):
    dim, n, w, per = _multigrid_mesh(seed)
    rng = np.random.default_rng(seed)
    pre, post = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    N = int(np.prod(n))
    b = rng.uniform(-1, 1, N)
    b -= b.mean()
    forms = ((0, 0, 0), (1, 0, 0), (1, 1024, 0), (1, 1024, 1), (0, 1024, 1), (1, 4096, 1), (1, 200, 1))
    ref = None
    for fuse, tail, lds in forms:
        poison()
        s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_fuse_small_levels={fuse}\npib_coarse_tail={tail}\n"
                                                                                 f"pib_coarse_tail_lds={lds}\n"))
        if any(per):
            s.setPeriodic(per)
        s.assemblePoisson(n, w, 0.01, capi.NULLSPACE_CONSTANT)
        first = None
        for rep in range(reps):
            x = np.zeros(N)
            poison_lds()
            s.solve(x, b)
            if first is None:
                first = x.copy()
                if ref is None:
                    ref = first
                elif not np.array_equal(first, ref):
                    bad += 1
                    print(f"seed {seed} n {n} per {per} V({pre},{post}) form {(fuse, tail, lds)}: differs from the per-phase form, max {np.abs(first - ref).max():.3e}", flush=True)
            elif not np.array_equal(x, first):
                bad += 1
                d = np.abs(x - first)
                print(f"seed {seed} n {n} per {per} V({pre},{post}) form {(fuse, tail, lds)} rep {rep}: differs from its own first run, max {d.max():.3e} at {int(d.argmax())}, {int((d > 0).sum())} entries", flush=True)
        s.destroy()
print("mismatches:", bad)
