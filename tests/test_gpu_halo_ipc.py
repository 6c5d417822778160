"""Device-initiated ghost refresh between two PROCESSES: each maps the other's inbox with
hipIpcOpenMemHandle, producers store into it and bump its counters, consumers wait on their own counters.  Both
processes use GPU 0 -- the cross-process, IPC-mapped variant of what the emulated-rank tests run inside one address
space; xGMI itself needs a multi-GPU box (tests/test_gpu_multi.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_refresh_ghosts_through_ipc_inboxes():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("FEMO_HALO_RCCL", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_halo_ipc_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, (o[-2000:], e[-4000:])
        assert "halo over hipIpc rank" in o
