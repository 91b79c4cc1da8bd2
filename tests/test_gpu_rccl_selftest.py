"""The RCCL calls of the production transport on hardware.  RCCL refuses several ranks per device and the
GPU box has one GPU: the multi-rank ALGORITHM runs through the loopback transport (test_gpu_multirank_loopback.py) and
through gloo on CPU (test_distributed_gloo.py); what is left to cover is that halo.hip's RCCL calls themselves are valid
-- run here in a one-rank RCCL world (pib_comm_selftest) -- and that bench.py starts under the driver's launcher."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.parametrize("n_owned,ghost", [(4096, 64), (3 * 512 * 512, 512 * 512), (10, 4)])
def test_rccl_entry_points_in_a_one_rank_world(n_owned, ghost):
    """grouped ncclSend / ncclRecv (wall-bounded: an empty group; periodic ring: both neighbours are the rank itself, two
    messages to one peer matched in issue order -- the P == 2 case of halo_exchange_planes -- and the same ring for the
    segmented plan of the packed velocity ordering: several pieces each way), the same exchange on the
    communication stream between two events, ncclAllReduce in place (PIB_NRED scalars, a large buffer), ncclAllGather in
    place and the grouped-ncclBroadcast form: everything that arrives is what must arrive."""
    from petibm_amd import capi
    lib = capi.load()
    err, cnt = C.c_double(-1.0), C.c_int(0)
    capi.check(lib.pib_comm_selftest(0, n_owned, ghost, C.byref(err), C.byref(cnt)))
    assert cnt.value == 1
    assert err.value == 0.0


def test_comm_latency_in_its_own_one_rank_world():
    """pib_comm_latency(NULL, ...) builds its own one-rank RCCL world (tools/comm_latency.py): ONE holder adopts the
    communicator (a second adoption destroyed it under the exchanges -- round-5 advisor finding); both timings come back
    positive and a second call in the same process works (the first one's communicator was destroyed exactly once)."""
    from petibm_amd import capi
    lib = capi.load()
    for _ in range(2):
        us = (C.c_double * 2)(-1.0, -1.0)
        capi.check(lib.pib_comm_latency(None, 1024, 5, us))
        assert us[0] > 0.0 and us[1] > 0.0


def test_selftest_rejects_a_halo_wider_than_the_slab():
    from petibm_amd import capi
    lib = capi.load()
    err, cnt = C.c_double(0.0), C.c_int(0)
    assert lib.pib_comm_selftest(0, 6, 4, C.byref(err), C.byref(cnt)) != 0
    assert b"selftest" in lib.pib_last_error()


def test_bench_under_the_drivers_launcher_with_one_rank():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ... bench.py --gpus 1`: the
    way the driver starts every N (RANK / LOCAL_RANK / WORLD_SIZE from the environment); one JSON line, n_gpus 1."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid", "128", "--steps", "1", "--warmup", "1",
           "--no-cpu", "--no-secondary", "--kernel-reps", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["counters"]["comm_ranks"] == 1
    assert d["true_rel_residual"] <= 1.5e-10


def test_bench_under_the_drivers_launcher_at_256_cubed():
    """the same launcher line on BASELINE config 2's grid (`--gpus 1 --grid 256`): the fused marches of the 256^3 level, the
    roofline entry and the same-run PMC switch all come up under torch.distributed.run"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid", "256", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--no-secondary", "--kernel-reps", "2", "--pmc", "off"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["iters_per_solve"] == 11 and d["true_rel_residual"] <= 1.5e-10
    assert d["config"]["grid"] == [256, 256, 256] and d["config"]["transport"] == "none" and d["roofline"]["frac"] > 0.3
