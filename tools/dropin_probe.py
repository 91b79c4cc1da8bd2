#!/usr/bin/env python3
"""The unchanged-PetIBM route at size on one GPU (round 5): the pinned-row convention on the device-assembled operator with the
fusions on / off, and the AmgX plug point end to end (host CSR through setMatrix only, host x / b), pageable and page-locked.

    python tools/dropin_probe.py [n]            (default 512)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime in the process: see petibm_amd/capi.py)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dt = 5e-4 if n == 512 else 1e-3
base = bench.solver_config("gmg", 1e-10, 1000, 0.9, 2, 2)
for name, extra in (("pinned, fusions on", ""), ("pinned, round-4 exclusions (pib_pin_sum_local=0, no post pair / residual-restriction march)",
                                                 "pib_pin_sum_local=0\npib_fuse_post_pair=0\npib_fuse_residual_restrict=0\n")):
    r = bench.pinned_device_case(n, dt, base + extra, steps=3)
    print(name, json.dumps(r), flush=True)
from petibm_amd import capi  # noqa: E402
from petibm_amd.linsolver import LinSolverHIP  # noqa: E402
import numpy as np  # noqa: E402
r, _, _, _ = bench.poisson_case(n, dt, base, "cosine", 3, 1, 5)
print("constant null space (the headline's convention)", json.dumps(r), flush=True)
for reg in (False, True):
    print("drop-in route, host buffers", "page-locked" if reg else "pageable", json.dumps(bench.dropin_amgx_route_case(n, dt, 1e-10, reg, steps=3)), flush=True)
