"""The model communicator (femo_comm_model, include/femo_hip_test.h): ONE context runs the N-rank code paths of the
library alone on its GPU -- collectives counted, nothing moved.  bench.py's scaling model times a rank's iteration this
way; here the path is checked for what it computes: the rank's block with zero ghost values, i.e. the principal
submatrix problem on its owned rows, which a direct solve reproduces."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from femo_amd.dist.partition import build_local_mesh, rcb_partition
from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,rank", [(8, 0), (8, 5), (2, 1)])
def test_model_communicator_runs_the_n_rank_merged_loop(world, rank):
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(48, 0.2)
    part = rcb_partition(m.x, world)
    L = build_local_mesh(m.x, m.conn, part, rank, world)
    ctx = E.Context(0)
    ctx.comm_model(rank, world)
    dm = E.DeviceMesh(ctx, L.x, L.conn, n_rows=L.n_owned)
    dm.set_global(m.x.min(axis=0), m.x.max(axis=0), m.n_vert)
    dm.set_halo(L.nbr, L.send_ptr, L.send_idx, L.recv_ptr)
    rng = np.random.default_rng(7)
    f = 1.0 + rng.random(len(L.conn))
    bd = fo.boundary_vertices_box(L.x)                     # box boundary of the WHOLE mesh among the local vertices (owned and ghost)
    bc = E.DirichletSet(dm, bd, np.zeros(len(bd)))
    nloc = len(L.x)
    A, b = E.Mat(dm), E.Vec(ctx, L.n_owned)
    E.assemble_system(dm, 0, None, E.Vec(ctx, nloc).fill(0.0), E.Vec(ctx, len(L.conn)).set(f), bc, None, A, b)
    x = E.Vec(ctx, nloc)
    A.solve_cg(b, x, rtol=1e-11, pc="bpx")
    ctx.sync()
    ctx.comm_stats(reset=True)
    info = A.solve_cg(b, x, rtol=1e-11, pc="bpx")
    st = ctx.comm_stats()
    assert info.converged == 1 and dm.pc_info()["levels"] >= 4
    # round 6: the model rank refreshes its ghosts the device-initiated way against its own scratch (loopback): the same
    # producer stores, counter bumps and consumer waits as a real rank, and no consumer ever gave up waiting
    hd = dm.halo_direct_info()
    assert hd["enabled"] == 1 and hd["timeouts"] == 0 and hd["exchanges"] >= st["neighbor_calls"]
    # the merged loop: one all-reduce and one neighbour exchange per enqueued iteration (+ the set-up reduction and the first apply)
    enqueued = st["neighbor_calls"] - 1                       # (the first application of the preconditioner starts an exchange too)
    assert info.loop_allreduces == enqueued
    assert info.iterations <= enqueued < info.iterations + 8
    assert st["allreduce_calls"] == enqueued + 2
    # what it solved: the owned-row block of the local operator (ghost values are never refreshed: they stay zero)
    om = fo.OMesh(3, np.ascontiguousarray(L.x), np.ascontiguousarray(L.conn))
    K = fo.eliminate_bc(fo.stiffness(om).tocsr(), bd).tocsr()
    n = L.n_owned
    ref = spla.spsolve(sp.csc_matrix(K[:n, :n]), b.get(n))
    xs = x.get(n)
    assert np.abs(xs - ref).max() <= 1e-8 * np.abs(ref).max()
