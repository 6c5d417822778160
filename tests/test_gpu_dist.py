"""What of the multi-GPU path can be exercised on ONE GPU: a 1-rank RCCL communicator
(init, all-reduce, the fold + all-reduce CG code path) and the ghost-DOF halo exchange
through ncclSend/ncclRecv to self on a mesh whose ghosts duplicate owned vertices."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm_ctx():
    from femo_amd import _lib
    from femo_amd.engine import Context
    if _lib.device_count() < 1:
        pytest.fail("no HIP device")
    c = Context(0)
    c.comm_init(Context.comm_unique_id(), 0, 1)
    yield c


def test_single_rank_communicator(comm_ctx):
    assert np.allclose(comm_ctx.allreduce_sum([1.5, -2.0, 3.25]), [1.5, -2.0, 3.25])


def test_single_reduction_cg_matches_standard_cg(comm_ctx, monkeypatch):
    """nranks > 1 uses the Chronopoulos-Gear recurrence (one fold + all-reduce per iteration);
    FEMO_FORCE_MULTI runs it on the 1-rank communicator."""
    from femo_amd import engine as E
    import scipy.sparse.linalg as spla
    for d, n in [(3, 20), (2, 150)]:
        m = fo.unit_cube_mesh(n, 0.2) if d == 3 else fo.unit_square_mesh(n, 0.2)
        dm = E.DeviceMesh(comm_ctx, m.x, m.conn)
        bd = fo.boundary_vertices_box(m.x)
        A = E.Mat(dm)
        E.assemble_jacobian(dm, 0, None, None, None, E.DirichletSet(dm, bd, 0.0), A)
        b = np.random.default_rng(1).standard_normal(m.n_vert)
        B, X1, X2 = E.Vec(comm_ctx, m.n_vert).set(b), E.Vec(comm_ctx, m.n_vert), E.Vec(comm_ctx, m.n_vert)
        monkeypatch.delenv("FEMO_FORCE_MULTI", raising=False)
        i1 = A.solve_cg(B, X1, rtol=1e-13)
        monkeypatch.setenv("FEMO_FORCE_MULTI", "1")
        i2 = A.solve_cg(B, X2, rtol=1e-13)
        i3 = A.solve_cg(B, X1, rtol=1e-13, zero_guess=False)       # warm start from the solution
        monkeypatch.delenv("FEMO_FORCE_MULTI")
        assert i1.converged == 1 and i2.converged == 1 and abs(i1.iterations - i2.iterations) <= 3
        xo = spla.splu(fo.eliminate_bc(fo.stiffness(m), bd).tocsc()).solve(b)
        assert np.abs(X2.get() - xo).max() < 1e-10 * np.abs(xo).max()
        assert i3.converged == 1 and i3.iterations <= 2


@pytest.mark.parametrize("n", [9, 26])
def test_halo_exchange_to_self(comm_ctx, n):
    """Ghost vertices duplicating owned ones, refreshed by ncclSend/ncclRecv to rank 0 itself.
    SpMV runs the overlapped path: interior slices while the halo is in flight on the second
    stream, slices with ghost columns after it (n=26: 309 slices, most of them interior)."""
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(n, 0.2)
    rng = np.random.default_rng(3)
    nv = m.n_vert
    dup = np.sort(rng.choice(nv, size=40, replace=False)).astype(np.int32)
    x_ext = np.vstack([m.x, m.x[dup]])
    conn = m.conn.copy()
    ghost_id = {int(v): nv + k for k, v in enumerate(dup)}
    for c in rng.choice(m.n_cell, size=m.n_cell // 3, replace=False):      # reroute some references to the copies
        for a in range(4):
            if int(conn[c, a]) in ghost_id and rng.random() < 0.7:
                conn[c, a] = ghost_id[int(conn[c, a])]
    used = np.unique(conn)
    assert np.all(np.isin(np.arange(nv, nv + len(dup)), used))            # every ghost referenced
    dm = E.DeviceMesh(comm_ctx, x_ext, conn, n_rows=nv)
    dm.set_halo([0], [0, len(dup)], dup, [0, len(dup)])
    ext = fo.OMesh(3, x_ext, conn)
    K = fo.stiffness(ext)[:nv]
    u, f = rng.standard_normal(nv), rng.standard_normal(m.n_cell)
    u_ext = np.concatenate([u, u[dup]])
    U = E.Vec(comm_ctx, nv + len(dup)).set(np.concatenate([u, np.full(len(dup), 1e30)]))   # ghosts hold garbage
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, None, J)
    Y = E.Vec(comm_ctx, nv + len(dup))
    for rep in range(3):                                                     # repeated: stream/event reuse
        U.set(np.concatenate([u * (rep + 1), np.full(len(dup), 1e30)]))
        J.mult(U, Y)                                                         # halo exchange overlapped with SpMV
        assert np.array_equal(U.get()[nv:], (rep + 1) * u[dup])
        assert np.abs(Y.get(nv) - (rep + 1) * (K @ u_ext)).max() < 1e-12 * (rep + 1) * np.abs(K @ u_ext).max()
    U.set(np.concatenate([u, np.zeros(len(dup))]))
    F, R = E.Vec(comm_ctx, m.n_cell).set(f), E.Vec(comm_ctx, nv)
    E.assemble_residual(dm, 0, None, U, F, R)
    ref = fo.residual(ext, u_ext, f)[:nv]
    assert np.abs(R.get() - ref).max() < 1e-12 * np.abs(ref).max()


def test_operator_stack_on_a_partitioned_mesh(comm_ctx):
    """The SPMD path of bench.py --gpus N with one rank: DistMesh (owned rows + halo plan) through
    FEA / FEAModel / Simulator, single-reduction CG, all-reduced dots; checked against the oracle."""
    import bench as B
    from femo_amd.dist import partition_mesh
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitCubeMesh
    utils_hip.set_context(comm_ctx)
    utils_hip.clear_workspaces()
    n = 12
    gmesh = createUnitCubeMesh(n, jitter=0.2)
    mesh = partition_mesh(gmesh, 0, 1)
    assert mesh.n_owned == gmesh.n_vert and len(mesh.local.nbr) == 0 and mesh.local.cell_owned.all()
    sim, fea = B.build_problem(mesh, device=True)
    f = B.source_fields(mesh, 1)[0]
    g = np.asarray(B.one_cycle(sim, fea, f))
    om = fo.unit_cube_mesh(n, jitter=0.2)
    bd = fo.boundary_vertices_box(om.x)
    ref = fo.reference_cycle(om, f, fo.u_target(om.x), bd, np.zeros(len(bd)))
    assert np.abs(sim['u'] - ref['u']).max() < 1e-10 * np.abs(ref['u']).max()
    assert np.abs(g - ref['grad']).max() < 1e-10 * np.abs(ref['grad']).max()
    utils_hip.clear_workspaces()


@pytest.mark.parametrize("pc", ["jacobi", "bpx"])
def test_solve_on_self_halo_mesh(comm_ctx, monkeypatch, pc):
    """The Krylov loops with a live halo plan (ghosts duplicating owned vertices, exchanged by
    ncclSend/ncclRecv to self): overlapped SpMV inside the loop, all-reduced scalars and (bpx) the
    all-reduced lattice accumulators.  Only Dirichlet vertices are duplicated: their rows are
    replaced by identity rows anyway, so the eliminated operator is exactly the plain mesh's SPD one
    (an owned row whose own cells point at a ghost copy would lose those cells' contributions)."""
    from femo_amd import engine as E
    import scipy.sparse.linalg as spla
    m = fo.unit_cube_mesh(14, 0.2)
    rng = np.random.default_rng(9)
    nv = m.n_vert
    dup = np.sort(rng.choice(fo.boundary_vertices_box(m.x), size=120, replace=False)).astype(np.int32)
    x_ext = np.vstack([m.x, m.x[dup]])
    conn = m.conn.copy()
    ghost_id = {int(v): nv + k for k, v in enumerate(dup)}
    for c in rng.choice(m.n_cell, size=m.n_cell // 2, replace=False):
        for a in range(4):
            if int(conn[c, a]) in ghost_id and rng.random() < 0.7:
                conn[c, a] = ghost_id[int(conn[c, a])]
    dm = E.DeviceMesh(comm_ctx, x_ext, conn, n_rows=nv)
    dm.set_halo([0], [0, len(dup)], dup, [0, len(dup)])
    bd_ext = fo.boundary_vertices_box(x_ext)
    bc = E.DirichletSet(dm, bd_ext, 0.25)
    f = 1.0 + rng.random(m.n_cell)
    n_ext = nv + len(dup)
    A, b = E.Mat(dm), E.Vec(comm_ctx, nv)
    E.assemble_system(dm, 0, None, E.Vec(comm_ctx, n_ext).fill(0.0), E.Vec(comm_ctx, m.n_cell).set(f), bc, None, A, b)
    # reference: the plain mesh, Newton right-hand side at u = 0 with g = 0.25 on the boundary
    bd = fo.boundary_vertices_box(m.x)
    K = fo.stiffness(m).tocsr()
    g = np.zeros(nv)
    g[bd] = 0.25
    rhs = -fo.load_vector(m, f) - K @ (-g)          # F(0) - K[:,bc](u - g) with u = 0
    rhs[bd] = -0.25                                  # rows of the set: u - g
    x_ref = spla.spsolve(fo.eliminate_bc(K, bd).tocsc(), rhs)
    assert np.abs(b.get() - rhs).max() < 1e-12 * np.abs(rhs).max()
    for force in (False, True):
        if force:
            monkeypatch.setenv("FEMO_FORCE_MULTI", "1")
        else:
            monkeypatch.delenv("FEMO_FORCE_MULTI", raising=False)
        x = E.Vec(comm_ctx, n_ext)
        info = A.solve_cg(b, x, rtol=1e-14, pc=pc)
        assert info.converged == 1
        assert np.abs(x.get(nv) - x_ref).max() < 1e-10 * np.abs(x_ref).max()
    monkeypatch.delenv("FEMO_FORCE_MULTI", raising=False)
