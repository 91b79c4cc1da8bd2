"""tools/destroy_stress.py [cycles] [procs]: create / assemble / solve / destroy small 3-D multigrid solvers in a loop, in `procs`
fresh processes one after the other -- the hunt for the intermittent crash inside pib_destroy (docs/history/round4.md): prints how
many of the processes died and with what."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))


def child(cycles):
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_fuzz import _multigrid_mesh
    from test_gpu_parity import gmg_cfg
    forms = ((0, 0, 0), (1, 0, 0), (1, 1024, 0), (1, 1024, 1), (0, 1024, 1), (1, 4096, 1), (1, 200, 1))
    if os.environ.get("STRESS_THREADS"):
        # solvers created, run (stream capture in thread-local mode) and destroyed on threads that then EXIT -- what the loopback
        # tests do with their rank threads -- before the main thread's loop
        import threading

        def one(seed):
            dim, n, w, per = _multigrid_mesh(seed)
            N = int(np.prod(n))
            b = np.random.default_rng(seed).uniform(-1, 1, N)
            b -= b.mean()
            s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=1))
            if any(per):
                s.setPeriodic(per)
            s.assemblePoisson(n, w, 0.01, capi.NULLSPACE_CONSTANT)
            x = np.zeros(N)
            s.solve(x, b)
            s.destroy()

        for rnd in range(int(os.environ["STRESS_THREADS"])):
            th = [threading.Thread(target=one, args=(2 * t,)) for t in range(4)]
            for t in th:
                t.start()
            for t in th:
                t.join()
    done = 0
    seed = 0
    while done < cycles:
        dim, n, w, per = _multigrid_mesh(seed)
        seed = (seed + 2) % 16  # the 3-D ones
        rng = np.random.default_rng(seed)
        N = int(np.prod(n))
        b = rng.uniform(-1, 1, N)
        b -= b.mean()
        for fuse, tail, lds in forms:
            s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=2, post=1, extra=f"pib_fuse_small_levels={fuse}\npib_coarse_tail={tail}\n"
                                                                                 f"pib_coarse_tail_lds={lds}\n"))
            if any(per):
                s.setPeriodic(per)
            s.assemblePoisson(n, w, 0.01, capi.NULLSPACE_CONSTANT)
            x = np.zeros(N)
            s.solve(x, b)
            s.destroy()
            done += 1
    print("child ok", done, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bad = 0
    for p in range(procs):
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "--child", str(cycles)], capture_output=True, text=True)
        tail = [l for l in r.stderr.splitlines() if "Fatal" in l or "what()" in l or "terminate" in l]
        print(f"process {p}: rc {r.returncode} {r.stdout.strip()} {' | '.join(tail)[:200]}", flush=True)
        bad += r.returncode != 0
    print(f"{bad} of {procs} processes died (mode {os.environ.get('PIB_DESTROY_MODE', 'default')})")
