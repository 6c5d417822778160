"""Worker of tests/test_gpu_halo_ipc.py: one of two PROCESSES that share GPU 0 and refresh each other's ghosts through
hipIpc-mapped inboxes (device-initiated ghost refresh, include/femo_hip.h ABI 9).  gloo is the control plane; the
context runs under the model communicator (no RCCL: it cannot place two ranks on one device), which leaves exactly the
neighbour exchange to be real."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from femo_amd import engine as E
    from femo_amd.dist import TorchControl, connect_halo_direct
    from femo_amd.dist.partition import build_local_mesh, rcb_partition
    from oracle import femo_oracle as fo
    control = TorchControl(rank, world)
    ctx = E.Context(0)
    ctx.comm_model(rank, world)
    ctx.control = control
    m = fo.unit_cube_mesh(14, 0.2)
    part = rcb_partition(m.x, world)
    L = build_local_mesh(m.x, m.conn, part, rank, world)
    dm = E.DeviceMesh(ctx, L.x, L.conn, n_rows=L.n_owned)
    dm.set_halo(L.nbr, L.send_ptr, L.send_idx, L.recv_ptr)
    assert dm.halo_direct_info()["enabled"] == 1                 # the model communicator's loopback plan ...
    assert connect_halo_direct(control, dm, L), "IPC plan refused"   # ... replaced by the real one: exported, opened, self-tested on both sides
    info = dm.halo_direct_info()
    assert info["enabled"] == 1 and info["timeouts"] == 0 and info["exchanges"] == 3
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, None, J)
    K = fo.stiffness(m).tocsr()
    own = L.vert_global[:L.n_owned]
    worst = 0.0
    for rep in range(6):                                         # both generations of the inbox, several epochs
        u = np.random.default_rng(100 + rep).standard_normal(m.n_vert)
        ul = np.full(len(L.x), 1e30)                             # ghosts hold garbage until the neighbour's stores arrive
        ul[:L.n_owned] = u[own]
        U, Y = E.Vec(ctx, len(L.x)).set(ul), E.Vec(ctx, len(L.x))
        if rep % 2 == rank:
            time.sleep(0.25)                                     # skew: the other process's consumer really WAITS on its counters
        J.mult(U, Y)
        ref = (K @ u)[own]
        worst = max(worst, float(np.abs(Y.get(L.n_owned) - ref).max() / np.abs(ref).max()))
        ghosts = U.get()[L.n_owned:]
        assert np.array_equal(ghosts, u[L.vert_global[L.n_owned:]])          # the neighbour's values, bit for bit
    info = dm.halo_direct_info()
    assert worst < 1e-13, worst
    assert info["timeouts"] == 0 and info["exchanges"] == 9, info
    control.barrier()
    print(f"halo over hipIpc rank {rank}: {len(L.x) - L.n_owned} ghosts, 6 products, worst error {worst:.1e}", flush=True)


if __name__ == "__main__":
    main()
