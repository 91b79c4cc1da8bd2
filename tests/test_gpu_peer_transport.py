"""The peer transport (csrc/halo.hip, pib_comm_peer_id): ONE PROCESS PER RANK, the neighbours' vectors mapped through HIP
IPC and pulled with device-to-device copies ordered by interprocess events, the ranks meeting in a POSIX shared-memory
segment.  RCCL refuses several ranks per device; this transport does not, so here the multi-PROCESS path of SURVEY.md 8e
-- the id made by rank 0 and handed to every rank, the attach, z-slab assembly, halo plans incl. the periodic ring and the
packed velocity ordering, all-reduced recurrences, distributed / replicated multigrid levels, the time step and immersed
bodies on slabs, and bench.py's own N > 1 launch -- runs end to end on the one test GPU, P processes at a time
(tests/peer_worker.py is one rank)."""
import ctypes
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_multirank_loopback import _cfg
from test_gpu_parity import rhs_for

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = os.path.join(ROOT, "tests", "peer_worker.py")


def peer_id(order=None) -> bytes:
    """order None: the default (device-ordered unless PIB_PEER_ORDER=host), "host" / "device": that ordering"""
    from petibm_amd import capi
    uid = ctypes.create_string_buffer(capi.UID_BYTES)
    if order is None:
        capi.check(capi.load().pib_comm_peer_id(uid))
    else:
        capi.check(capi.load().pib_comm_peer_id_ordered(uid, 1 if order == "device" else 0))
    assert uid.raw[:8] == b"PIBPEER1"
    return uid.raw


def run_ranks(tmp_path, job, P, timeout=420, order=None):
    """P worker processes on the one GPU; returns their result files"""
    job = dict(job, P=P, uid=peer_id(order))
    path = os.path.join(tmp_path, "job.pkl")
    pickle.dump(job, open(path, "wb"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PIB_PEER_TIMEOUT_S="240")
    procs = [subprocess.Popen([sys.executable, WORKER, path, str(r)], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(P)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(P)]


@pytest.mark.parametrize("P,n,per,pc,extra,sweeps", [
    (2, (16, 16, 16), None, "BLOCK_JACOBI", "", 1),
    (3, (24, 20), None, "BLOCK_JACOBI", "", 1),                                        # 2-D: slabs along y
    (2, (16, 16, 32), None, "AMG", "pib_agglomerate_below=10\n", 1),                    # several distributed levels
    (4, (32, 32, 32), None, "AMG", "pib_agglomerate_below=100\n", 2),                   # the bench's V(2,2) cycle
    (3, (128, 16, 96), None, "AMG", "pib_march_min_cells=0\npib_agglomerate_below=100\npib_overlap_min_bytes=0\n", 2),  # fused kernels, overlapped exchange
    (2, (16, 16, 32), (True, True, True), "AMG", "", 1),                               # periodic slab axis: the ring, P = 2
    (3, (16, 16, 36), (True, False, True), "AMG", "", 1),
])
def test_poisson_solve_across_processes_matches_single_rank(tmp_path, P, n, per, pc, extra, sweeps):
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    dt = 0.01
    m = omesh.create_mesh(omesh.periodic_config(n, per) if per else omesh.uniform_config(n))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    w = [m.dL[3][d].true for d in range(m.dim)]
    cfg = _cfg(pc, extra=extra, sweeps=sweeps)
    res = run_ranks(str(tmp_path), dict(kind="poisson", n=n, w=w, dt=dt, cfg=cfg, xs=xs, b=b, periodic=per), P)
    y = np.concatenate([r["y"] for r in res])
    x = np.concatenate([r["x"] for r in res])
    if per and per[-1]:
        assert np.abs(y - b).max() <= 4e-16 * np.abs(A.val).max() * np.abs(xs).max() * 8  # wrapped neighbour summed first
    else:
        assert np.array_equal(y, b)  # SpMV across the slab boundaries: bit-identical to the oracle
    assert len({int(r["its"]) for r in res}) == 1
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    assert all(int(r["counters"][5]) == P for r in res)  # ranks the transport reports
    s1 = LinSolverHIP("poisson", config_text=cfg)
    if per:
        s1.setPeriodic(per)
    s1.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert int(res[0]["its"]) <= s1.getIters() + (6 if per else 1)
    e = (x - x.mean()) - (x1 - x1.mean())
    assert np.linalg.norm(e) <= 1e-7 * np.linalg.norm(x1)
    s1.destroy()


@pytest.mark.parametrize("P,n,pc,extra,sweeps", [
    (4, (32, 32, 32), "AMG", "pib_agglomerate_below=100\n", 2),
    (3, (128, 16, 96), "AMG", "pib_march_min_cells=0\npib_agglomerate_below=100\npib_overlap_min_bytes=0\n", 2),
    (2, (24, 20), "BLOCK_JACOBI", "", 1),
])
def test_device_ordered_collectives_give_the_host_ordered_bits(tmp_path, P, n, pc, extra, sweeps):
    """The two orderings of the peer transport move the same data and sum the scalars in the same (rank) order: products,
    iterates, residual histories and solutions agree bit for bit."""
    dt = 0.01
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    w = [m.dL[3][d].true for d in range(m.dim)]
    job = dict(kind="poisson", n=n, w=w, dt=dt, cfg=_cfg(pc, extra=extra, sweeps=sweeps), xs=xs, b=b, periodic=None)
    out = {}
    for order in ("host", "device"):
        d = os.path.join(str(tmp_path), order)
        os.makedirs(d)
        out[order] = run_ranks(d, job, P, order=order)
    for h, dv in zip(out["host"], out["device"]):
        assert int(h["its"]) == int(dv["its"])
        assert np.array_equal(h["y"], dv["y"]) and np.array_equal(h["x"], dv["x"]) and np.array_equal(h["hist"], dv["hist"])
        assert np.array_equal(h["counters"][:6], dv["counters"][:6])


@pytest.mark.parametrize("P,n,pc", [(2, (96, 96), "BLOCK_JACOBI"), (3, (24, 24, 24), "AMG")])
def test_iteration_graph_on_ranks(tmp_path, P, n, pc):
    """Launch-bound systems on several ranks: with the device-ordered transport the iteration body -- kernels, plane
    exchanges, all-reduces, the all-gather at the replicated-level switch -- is captured once and replayed (the collectives
    count themselves on the device, so a replay issues the right numbers).  Same bits as plain launches, and the replays
    are really taken."""
    dt = 0.01
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, 0.005)
    xs, b = rhs_for(A)
    w = [m.dL[3][d].true for d in range(m.dim)]
    out = {}
    for g in (0, 1):
        d = os.path.join(str(tmp_path), f"graph{g}")
        os.makedirs(d)
        job = dict(kind="poisson", n=n, w=w, dt=dt, cfg=_cfg(pc, extra=f"pib_graph_max_rows={4194304 if g else 0}\npib_agglomerate_below=100\n"), xs=xs, b=b,
                   periodic=None, timed_solves=3)
        out[g] = run_ranks(d, job, P, order="device")
    for a, c in zip(out[0], out[1]):
        assert int(a["graph_replays"]) == 0 and int(c["graph_replays"]) >= int(c["its"]) - 2 > 0
        assert int(a["its"]) == int(c["its"]) and np.array_equal(a["x"], c["x"]) and np.array_equal(a["hist"], c["hist"])
    print(f"P={P} {n} {pc}: {1e3 * float(out[0][0]['seconds_per_solve']):.2f} ms per solve with plain launches, "
          f"{1e3 * float(out[1][0]['seconds_per_solve']):.2f} ms with the captured iteration ({int(out[1][0]['its'])} iterations)")


@pytest.mark.parametrize("case,P,bodies", [("3d_cavity", 2, False), ("2d_convective_outlet", 3, False), ("3d_sphere", 2, True),
                                           ("moving_cylinder", 3, True), ("3d_periodic_box", 2, False),
                                           ("3d_channel_periodic_z", 3, False)])
def test_time_step_across_processes_matches_single_rank(tmp_path, case, P, bodies):
    """the device time step (and the decoupled IBPM step with its all-reduced force system) on slabs, one process per rank"""
    import test_gpu_navierstokes_slabs as T
    from petibm_amd.navierstokes import DecoupledIBPMSolver, NavierStokesSolver
    steps = 3
    job = dict(kind="navierstokes", case=case, bodies=bodies, steps=steps)
    if bodies:
        cfg, bods, pose = T._ib_case(case)
        one = DecoupledIBPMSolver(cfg, bodies=bods, velocity_cfg=T.VEL, poisson_cfg=T.KSP_P, forces_cfg=T.FORCES)
        dt = cfg["parameters"]["dt"]
        for step in range(1, steps + 1):
            if pose is not None:
                x, v = pose(step * dt)
                one.moveBodies([x], [v])
            one.advance()
        f1 = np.asarray(one.getForces()[0])
    else:
        make, pinned = T.CASES[case]
        one = NavierStokesSolver(make(), velocity_cfg=T.VEL, poisson_cfg=T.AMGX_P if pinned else T.KSP_P)
        rng = np.random.default_rng(11)
        job["U0"] = 0.1 * rng.uniform(-1, 1, one.UN)
        job["p0"] = 0.1 * rng.uniform(-1, 1, one.pN)
        if "convective" in case:
            job["U0"][: int(np.prod(one._field_shape(0)))] += 1.0
        one.setState(job["U0"], job["p0"])
        one.advance(steps)
    job["U_ref"], job["p_ref"] = one.getState()
    one.destroy()
    res = run_ranks(str(tmp_path), job, P)
    for r in res:
        assert np.abs(r["U"] - r["U_ref"]).max() <= 1e-8 * max(1.0, np.abs(r["U_ref"]).max())
        if bodies:
            assert np.abs(r["forces"] - f1).max() <= 1e-7 * np.abs(f1).max()
            assert np.array_equal(r["forces"], res[0]["forces"])  # replicated force solves: the same bits on every rank


def test_bench_launches_its_own_ranks_over_the_peer_transport():
    """`python bench.py --gpus 2 --transport peer` with no launcher around it spawns its two ranks (torch.distributed.run on
    127.0.0.1), which share the one GPU here (PIB_BENCH_SHARE_GPU=1: torch side on gloo): the N > 1 bench path end to end
    in separate processes -- one JSON line, n_gpus 2, the transport's rank count, the residual contract."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PIB_BENCH_SHARE_GPU="1", PIB_PEER_TIMEOUT_S="240")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer", "--grid", "128", "--steps", "2",
           "--warmup", "1", "--no-cpu", "--no-secondary", "--kernel-reps", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["counters"]["comm_ranks"] == 2
    assert d["config"]["parallelism"] == "zslab2" and d["config"]["transport"] == "peer"
    assert d["true_rel_residual"] <= 1.5e-10 and d["counters"]["halo_exchanges"] > 0


@pytest.mark.parametrize("how", ["forced", "genuine"])
def test_bench_falls_back_to_the_peer_transport_when_rccl_fails_on_first_contact(how):
    """The first run on a multi-GPU node must end in a JSON line, not a traceback: with RCCL's bootstrap failing
    (PIB_FORCE_RCCL_FAIL=1 makes ncclCommInitRank's call site return its error; the ranks share the one GPU here, torch side on
    gloo) `bench.py --gpus 2` -- default transport rccl -- agrees on the failure across the ranks, switches to the peer
    transport, says so in `notes` and `config.transport`, and still meets the residual contract; the CG recurrence tuned on
    the untimed solves is named too, and every rank's counters are in the line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PIB_BENCH_SHARE_GPU="1", PIB_PEER_TIMEOUT_S="240")
    if how == "forced":
        env["PIB_FORCE_RCCL_FAIL"] = "1"
    # ("genuine": RCCL itself refuses two ranks on one device -- ncclCommInitRank returns `invalid usage` on every rank: a real
    # first-contact failure of the library, handled by the same path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PIB_TRANSPORT") + (("PIB_FORCE_RCCL_FAIL",) if how == "genuine" else ()):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "2", "--warmup", "1", "--no-cpu",
           "--no-secondary", "--kernel-reps", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["transport"] == "peer"
    assert any("fell back to --transport peer" in nt and ("PIB_FORCE_RCCL_FAIL" in nt if how == "forced" else "ncclCommInitRank" in nt)
               for nt in d["notes"]), d["notes"]
    assert any("CG recurrence tuned at first contact" in nt for nt in d["notes"]) and d["config"]["cg_recurrence"] in ("standard", "single-reduction")
    assert d["true_rel_residual"] <= 1.5e-10 and len(d["per_rank"]) == 2 and all(r["halo_exchanges"] > 0 for r in d["per_rank"])


def test_config3_512_cubed_on_8_processes():
    """BASELINE config 3 -- the 512^3 cavity on 8 z-slabs of 64 planes, multigrid-PCG V(2,2), rtol 1e-10 -- with EIGHT PROCESSES
    (bench.py's own launch, peer transport, all on the one GPU): the single-rank iteration count and the residual contract,
    every rank through the same ~110 collectives of a solve.  (test_config3_512_cubed_on_8_slabs runs the same on 8 loopback
    threads; timing means nothing here -- eight contexts time-slice one GPU.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PIB_BENCH_SHARE_GPU="1", PIB_PEER_TIMEOUT_S="300")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--transport", "peer", "--steps", "1", "--warmup", "0",
           "--no-cpu", "--no-secondary", "--kernel-reps", "1", "--no-tune-recurrence"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "zslab8" and d["counters"]["comm_ranks"] == 8
    assert d["iters_per_solve"] == 11 and d["true_rel_residual"] <= 1.5e-10


def test_a_missing_rank_is_an_error_not_a_hang(monkeypatch):
    """every wait for another rank is bounded (PIB_PEER_TIMEOUT_S): rank 0 of a two-rank world whose rank 1 never shows up gets
    the error code and a message that names the wait, and the shared-memory name does not outlive the attempt"""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    monkeypatch.setenv("PIB_PEER_TIMEOUT_S", "2")
    uid = peer_id()
    name = uid[8:].split(b"\0", 1)[0].decode()
    with pytest.raises(capi.PibError) as e:
        LinSolverHIP("poisson", config_text=_cfg("BLOCK_JACOBI"), rank=0, nranks=2, uid=uid, device=0)
    assert "peer transport" in str(e.value) and "waited" in str(e.value)
    assert not os.path.exists("/dev/shm" + name)
