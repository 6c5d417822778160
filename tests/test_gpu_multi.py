"""Two real ranks over RCCL (skipped on one-GPU boxes): `bench.py --gpus 2` under torchrun on a small cube --
halo exchange on the comm stream overlapped with interior rows, the lattice and scalar all-reduces on the main
stream, both over one ncclComm_t.  The emulated-rank tests cover the same library code with host-staged
collectives; these are the only tests that need xGMI.  Round 3: the partitioned shell over the same communicator."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("halo", ["direct", "rccl"])
def test_two_rank_bench_over_rccl(halo):
    """`halo`: the ghost refresh -- device-initiated over hipIpc-mapped inboxes (round 6; falls back collectively to
    ncclSend/Recv if its self-test fails on the box, which the record then shows) or ncclSend/Recv by request."""
    from femo_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FEMO_HALO_RCCL", None)
    if halo == "rccl":
        env["FEMO_HALO_RCCL"] = "1"
    # n = 48: the smallest cube whose lattice has the four levels the merged loop (one all-reduce per iteration) needs
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mesh-n", "48", "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 2 and c["n_dof"] == 49 ** 3 and sum(c["owned_per_rank"]) == 49 ** 3
    assert c["linear_solves_per_step"] == 4 and c["neighbours_per_rank"] == [1, 1]
    its = c["cg_iterations_per_step"]
    assert 10 < its[0] <= 40 and its[1] <= 2 and its[2] <= 2 and 10 < its[3] <= 40
    # the merged BPX-PCG over real RCCL: ncclSend/Recv on the comm stream concurrent with ONE ncclAllReduce per iteration on
    # the main stream (counted inside the solver loops); the run checks itself against the DST-exact cycle
    assert 1.0 <= c["allreduce_per_cg_iteration"] <= 1.3, c["allreduce_per_cg_iteration"]
    assert r["check"]["u_rel_err"] < 1e-10 and r["check"]["grad_rel_err"] < 1e-10
    hd = c["ghost_refresh_rank0"]
    assert hd["timeouts"] == 0 and (hd["enabled"] == 0 if halo == "rccl" else hd["enabled"] in (0, 1))
    if hd["enabled"]:
        assert hd["exchanges"] > sum(its)                # every iteration refreshed its ghosts through the inboxes


@pytest.mark.parametrize("world", [4, 8])
def test_rank_ladder_bench_over_rccl(world):
    """Round 6: the rest of the ladder -- `bench.py --gpus 4` and `--gpus 8` under torchrun, real
    RCCL over xGMI -- so that the first multi-GPU box runs 2, 4 and 8 ranks unattended.  n = 64: every rank of the 1 x 2 x 4
    pencils keeps >= 16 x-lines per cut direction and the whole mesh's lattice has the depth the merged loop needs.  The
    one-GPU iteration counts must be reproduced (the partition does not change the Krylov iteration), one all-reduce per
    iteration, and the run checks itself against the DST-exact cycle of the whole mesh."""
    from femo_amd import _lib
    if _lib.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--mesh-n", "64", "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == world and c["n_dof"] == 65 ** 3 and sum(c["owned_per_rank"]) == 65 ** 3
    assert len(c["owned_per_rank"]) == world and max(c["neighbours_per_rank"]) <= 8
    its = c["cg_iterations_per_step"]
    assert c["linear_solves_per_step"] == 4 and 10 < its[0] <= 40 and its[1] <= 2 and its[2] <= 2 and 10 < its[3] <= 40
    assert 1.0 <= c["allreduce_per_cg_iteration"] <= 1.3, c["allreduce_per_cg_iteration"]
    assert r["check"]["passed"] and r["check"]["u_rel_err"] < 1e-10 and r["check"]["grad_rel_err"] < 1e-10
    assert r["checks_passed"] is True


def test_two_rank_shell_over_rccl():
    """The partitioned shell (femo_shell_set_partition) with real ncclSend/ncclRecv + ncclAllReduce: tests/_shell_rccl_worker.py
    on two GPUs against the oracle's direct solve and exact adjoint gradient."""
    from femo_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "tests", "_shell_rccl_worker.py")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "shell over RCCL on 2 ranks" in p.stdout
