#!/usr/bin/env python3
"""The reference's examples/navierstokes/taylorgreenvortex3dRe1600_GPU on one MI355X: runs the case directory
(examples/cases/taylorgreenvortex3dRe1600: 256^3 cells, 2000 steps of dt = 0.01), samples the mean kinetic energy every
`--every` steps (the quantity of the reference's mean-kinetic-energy.png) and prints it next to the pseudo-spectral
512^3 data the reference ships (tests/golden/reference_test_vectors.json).

    python examples/python/taylor_green_3d.py [--cells 256] [--nt 2000] [--every 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from petibm_amd import navierstokes  # noqa: E402


def mean_kinetic_energy(s):
    U, _ = s.getState()
    e, off = 0.0, 0
    for _, shape in s._field_shapes()[: s.dim]:
        sz = int(np.prod(shape))
        e += 0.5 * float(np.mean(U[off:off + sz] ** 2))  # uniform periodic mesh: every point carries the same volume
        off += sz
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--nt", type=int, default=None)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--dt", type=float, default=None, help="time step (the case's 0.01 is the 256^3 one: CFL 0.4; halve it at 512^3)")
    ap.add_argument("--poisson-extra", default="", help="lines appended to the Poisson solver's configuration (';' separated)")
    ap.add_argument("--velocity-extra", default="", help="lines appended to the velocity solver's configuration (';' separated)")
    a = ap.parse_args()
    d = os.path.join(ROOT, "examples", "cases", "taylorgreenvortex3dRe1600")
    cfg = yaml.safe_load(open(os.path.join(d, "config.yaml")))
    if a.cells:
        for ax in cfg["mesh"]:
            ax["subDomains"][0]["cells"] = a.cells
    if a.dt:
        cfg["parameters"]["dt"] = a.dt
    nt = a.nt if a.nt is not None else int(cfg["parameters"]["nt"])
    texts = {k: open(os.path.join(d, cfg["parameters"][k]["config"])).read() for k in ("velocitySolver", "poissonSolver")}
    texts["poissonSolver"] += "".join(ln + "\n" for ln in a.poisson_extra.split(";") if ln)
    texts["velocitySolver"] += "".join(ln + "\n" for ln in a.velocity_extra.split(";") if ln)
    ref = {round(r[0], 6): r[1] for r in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")))[
        "taylor_green_vortex_3d_re1600_spectral_512"]["rows"]}
    t0 = time.perf_counter()
    s = navierstokes.NavierStokesSolver(cfg, velocity_cfg=texts["velocitySolver"], poisson_cfg=texts["poissonSolver"])
    print(f"{s.n} cells, {s.UN} velocity + {s.pN} pressure unknowns; set-up {time.perf_counter() - t0:.1f} s", flush=True)
    print("  step      t     E_k         spectral 512^3   v_its p_its", flush=True)
    wall = 0.0
    worst = 0.0
    done = 0
    while done < nt:
        k = min(a.every, nt - done)
        t1 = time.perf_counter()
        s.advance(k)
        ite, vi, vr, pi, pr = s.linSolversInfo()  # synchronises
        wall += time.perf_counter() - t1
        done += k
        e = mean_kinetic_energy(s)
        r = ref.get(round(s.t, 6))
        if r is not None:
            worst = max(worst, abs(e - r) / r)
        print(f"{done:6d} {s.t:6.2f} {e:.8f}  {'' if r is None else f'{r:.8f}':>14}   {vi:4d} {pi:4d}", flush=True)
    print(f"{nt} steps in {wall:.1f} s = {1e3 * wall / nt:.1f} ms/step; largest relative deviation of E_k from the "
          f"spectral data {worst:.3%}")
    s.destroy()


if __name__ == "__main__":
    main()
