"""One rank of the 2-GPU shell test (started by tests/test_gpu_multi.py under torch.distributed.run): the partitioned roof
solved over RCCL -- halo exchange of the node blocks by ncclSend/ncclRecv, all-reduced lattice residual, Galerkin blocks,
dense coarse operator and scalars -- against the oracle's direct solve.  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> int:
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    from femo_amd.dist import TorchControl
    from femo_amd.dist.shell import ShellPartition
    from femo_amd.engine import Context
    from femo_amd.fea.shell import ShellProblem, ShellSpace
    from oracle import shell_oracle as so
    from test_gpu_shell import E_ROOF, FZ, H_ROOF, roof_fixed

    control = TorchControl(rank, world)
    ctx = Context(local)
    control.init_comm(ctx)
    pts, conn = so.scordelis_lo_mesh(24, 20)
    V = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(0)
    h = H_ROOF * (1.0 + 0.3 * rng.random(V.n_vert))
    f = np.tile([0.0, 0.0, FZ], (V.n_vert, 1)) * (1.0 + 0.2 * rng.random((V.n_vert, 1)))
    fixed = roof_fixed(V)
    P = ShellPartition(ShellSpace(pts, conn), rank, world)
    prob = ShellProblem(pts, conn, E_ROOF, 0.3, fixed_dofs=fixed, ctx=ctx, partition=P)
    prob.set_thickness(h)
    prob.set_load(f)
    J, g, w = prob.compliance_gradient()
    its = prob.last_info.iterations
    # every rank checks its own points against the direct solve (the oracle runs on each rank: 10 k dofs)
    K = so.assemble(V, so.element_stiffness(V, h, E_ROOF, 0.3))
    wref = so.solve(K, so.load_vector(V, f), fixed)
    dJdw = so.compliance_du(V, wref)
    dJdw[fixed] = 0.0
    gref = -so.dform_dh(V, h, E_ROOF, 0.3, so.solve(K, dJdw, fixed), wref)
    ew = np.abs(w - wref[P.dof_global]).max() / np.abs(wref).max()
    ov = P.owned_vertices()
    eg = np.abs(g[ov] - gref[P.vert_global[ov]]).max() / np.abs(gref).max()
    eJ = abs(J - so.compliance(V, wref)) / abs(so.compliance(V, wref))
    worst = control.allreduce([ew, eg, eJ], "max")
    counts = control.gather([its])
    control.barrier()
    if rank == 0:
        print(f"shell over RCCL on {world} ranks: state {worst[0]:.2e}, gradient {worst[1]:.2e}, J {worst[2]:.2e}, iterations {counts.ravel()}")
    ok = worst[0] <= 1e-9 and worst[1] <= 1e-7 and worst[2] <= 1e-8 and len(set(counts.ravel().tolist())) == 1
    return 0 if ok else 1


if __name__ == "__main__":
    code = main()
    sys.stdout.flush()
    os._exit(code)
