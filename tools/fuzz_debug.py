"""Debug helper: the multigrid-forms fuzz case of one seed, every form twice, first history entries side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from petibm_amd import capi
from petibm_amd.linsolver import LinSolverHIP
from test_gpu_parity import gmg_cfg
from test_gpu_fuzz import _multigrid_mesh

for seed in [int(a) for a in sys.argv[1:]] or [12]:
    dim, n, w, per = _multigrid_mesh(seed)
    rng = np.random.default_rng(seed)
    pre, post = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    N = int(np.prod(n))
    b = rng.uniform(-1, 1, N)
    b -= b.mean()
    print("seed", seed, "dim", dim, "n", n, "per", per, "pre", pre, "post", post, flush=True)
    forms = ((0, 0, 0), (1, 0, 0), (1, 1024, 0), (1, 1024, 1), (0, 1024, 1), (1, 4096, 1), (1, 200, 1))
    ref = None
    for fuse, tail, lds in forms:
        for rep in range(2):
            s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_fuse_small_levels={fuse}\npib_coarse_tail={tail}\n"
                                                                                     f"pib_coarse_tail_lds={lds}\nsolv:max_iters=2\nsolv:error_if_not_converged=0\n"))
            if any(per):
                s.setPeriodic(per)
            s.assemblePoisson(n, w, 0.01, capi.NULLSPACE_CONSTANT)
            x = np.zeros(N)
            try:
                s.solve(x, b)
            except Exception as e:
                print("solve:", e)
            h = s.getResidualHistory().copy()
            if ref is None:
                ref = x.copy()
            d = np.abs(x - ref)
            print((fuse, tail, lds), rep, "hist", h[:3], "max|x-ref|", d.max(), "at", int(d.argmax()), "nnz diff", int((d > 0).sum()), flush=True)
            s.destroy()
