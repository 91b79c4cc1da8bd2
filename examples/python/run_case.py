#!/usr/bin/env python3
"""Run a simulation directory laid out like the reference's examples (config.yaml, config/*.info, body files) on the GPU.

    python examples/python/run_case.py examples/cases/cylinder2dRe40 [--nt N]

Picks the flow solver the way the reference's applications do (applications/navierstokes/main.cpp,
applications/decoupledibpm/main.cpp): immersed bodies -> decoupled IBPM, none -> Navier-Stokes; --app ibpm selects the
coupled IBPM of applications/ibpm.  Writes
output/iterations-<start>.txt (ite, iterations and residual per solver: navierstokes.cpp:766-794,
decoupledibpm.cpp:399-434) and, with bodies, output/forces-<start>.txt (t, fx, fy[, fz] per body:
decoupledibpm.cpp:437-465), output/grid.h5 and the solution / restart files output/<step>.h5 (every nsave / nrestart
steps; startStep > 0 restarts from output/<startStep>.h5) -- the files the reference's plotting scripts read."""
import argparse
import os
import sys
import time

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from petibm_amd import linsolver, navierstokes  # noqa: E402


def solver_text(cfg, key, directory):
    node = cfg["parameters"].get(key)
    if node is None:
        return None
    path = node.get("config", "None")
    if path == "None":
        return ""
    return open(path if os.path.isabs(path) else os.path.join(directory, path)).read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--nt", type=int, default=None, help="number of time steps (default: parameters.nt)")
    ap.add_argument("--vorticity", action="store_true",
                    help="append the vorticity fields of petibm-vorticity to every saved solution file (and their gridlines to grid.h5)")
    ap.add_argument("--xdmf", action="store_true", help="write the .xmf descriptions of petibm-createxdmf next to the HDF5 files")
    ap.add_argument("--app", default="auto", choices=["auto", "navierstokes", "decoupledibpm", "ibpm"],
                    help="which of the reference's applications to mirror (auto: decoupledibpm when the case has bodies)")
    ap.add_argument("--log-view", action="store_true",
                    help="time the stages of a step under the reference's PetscLogStage names (what `-log_view` prints per stage)")
    a = ap.parse_args()
    d = os.path.abspath(a.directory)
    cfg = yaml.safe_load(open(os.path.join(d, "config.yaml")))
    par = cfg["parameters"]
    nt = a.nt if a.nt is not None else int(par["nt"])
    start = int(par.get("startStep", 0))
    texts = {k: solver_text(cfg, k, d) for k in ("velocitySolver", "poissonSolver", "forcesSolver")}
    # an absent parameters.<name>Solver node means "the defaults" (linsolver.cpp:67: type CPU, config None)
    kw = {k: v for k, v in (("velocity_cfg", texts["velocitySolver"]), ("poisson_cfg", texts["poissonSolver"])) if v is not None}
    if a.app == "navierstokes":
        cfg = dict(cfg, bodies=None)
    if cfg.get("bodies"):
        cls = navierstokes.IBPMSolver if a.app == "ibpm" else navierstokes.DecoupledIBPMSolver  # applications/ibpm: coupled
        s = cls(cfg, forces_cfg=texts["forcesSolver"] or navierstokes.DEFAULT_FORCES_CFG, directory=d, **kw)
    else:
        s = navierstokes.NavierStokesSolver(cfg, **kw)
    out = os.path.join(d, "output")
    os.makedirs(out, exist_ok=True)
    nsave = int(par.get("nsave", 0))
    nrestart = int(par.get("nrestart", 0))
    try:
        from petibm_amd import h5io
        h5io.lib()
        have_h5 = True
    except ImportError as e:
        print(f"HDF5 output disabled: {e}")
        have_h5 = False
    if have_h5:
        s.writeGrid(os.path.join(out, "grid.h5"))                      # main.cpp: mesh->write(output/grid.h5)
        if start > 0:
            s.readRestartData(os.path.join(out, f"{start:07d}.h5"))    # ioInitialData (navierstokes.cpp:189-236)
        else:
            s.write(os.path.join(out, f"{start:07d}.h5"))
    mode = "a" if start > 0 else "w"
    it_file = open(os.path.join(out, f"iterations-{start}.txt"), mode)
    f_file = open(os.path.join(out, f"forces-{start}.txt"), mode) if cfg.get("bodies") else None
    if a.log_view:
        s.enableStageTimers()
    t0 = time.perf_counter()
    for _ in range(nt):
        s.advance()
        info = s.linSolversInfo()
        it_file.write("\t".join(f"{v:d}" if isinstance(v, int) else f"{v:e}" for v in info) + "\n")
        if f_file:
            _, avg = s.getForces()
            f_file.write(f"{s.t:10.8e}\t" + "\t".join(f"{v:10.8e}" for v in avg.reshape(-1)) + "\t\n")
        if have_h5 and nsave > 0 and s.ite % nsave == 0:
            s.write(os.path.join(out, f"{s.ite:07d}.h5"))
            if a.vorticity:
                s.writeVorticity(os.path.join(out, f"{s.ite:07d}.h5"), os.path.join(out, "grid.h5"))
        if have_h5 and nrestart > 0 and s.ite % nrestart == 0:
            s.writeRestartData(os.path.join(out, f"{s.ite:07d}.h5"))
    wall = time.perf_counter() - t0
    if a.xdmf and have_h5 and nsave > 0:
        from petibm_amd import xdmf
        xdmf.write_all(out, s.dim, s.n, s.periodic, range(start, start + nt + 1, nsave), vorticity=a.vorticity)
    it_file.close()
    if f_file:
        f_file.close()
    print(f"{nt} steps in {wall:.2f} s ({1e3 * wall / max(nt, 1):.2f} ms/step); last step: {s.linSolversInfo()}")
    if a.log_view:
        st = s.stageTimes()
        steps = max(st.pop("steps"), 1)
        total = sum(st.values())
        print(f"stage                 ms/step   share   ({steps} steps, HIP events on the engine's stream)")
        for name, ms in st.items():
            print(f"  {name:<18s} {ms / steps:9.3f}  {100.0 * ms / max(total, 1e-30):5.1f} %")
    s.destroy()


if __name__ == "__main__":
    main()
