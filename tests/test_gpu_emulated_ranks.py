"""The multi-rank code paths of libfemo_hip.so executed for real on ONE GPU: N contexts, one host
thread each, joined in a femo_emu_group whose collectives replace RCCL by host staging + barriers
(femo_amd/csrc/comm.cpp).  Each rank holds its RCB part of the mesh (owned rows first, ghosts behind),
its halo plan with REAL neighbours, and runs the same calls a torchrun job runs: assemble, halo
exchange, all-reduced dot products, Jacobi / BPX CG with the all-reduced lattice.  Results are checked
against the serial direct solve."""
import threading

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from femo_amd.dist.partition import build_local_mesh, rcb_partition
from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def _run_ranks(world, fn):
    """fn(rank, ctx) on `world` threads; returns the per-rank results, re-raises the first failure.  Every context carries a
    control plane (ThreadControl): meshes whose halo plan is installed through `_set_halo` / DistMesh.device() get the
    device-initiated ghost refresh (round 6) unless FEMO_HALO_RCCL=1 keeps the host-staged neighbour exchange."""
    from femo_amd.dist import ThreadControl
    from femo_amd.engine import Context, EmuGroup
    group = EmuGroup(world)
    shared = ThreadControl.Shared(world)
    out, err = [None] * world, [None] * world

    def body(rank):
        try:
            ctx = Context(0)
            ThreadControl(rank, shared, group).init_comm(ctx)
            out[rank] = fn(rank, ctx)
            ctx.sync()
        except BaseException as e:          # noqa: BLE001 - reported below, the other ranks time out on their own
            err[rank] = e

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for e in err:
        if e is not None:
            raise e
    return out


def _set_halo(ctx, dm, L):
    """The rank's halo plan + (collectively) the device-initiated refresh for it, as DistMesh.device() does."""
    from femo_amd.dist import connect_halo_direct
    dm.set_halo(L.nbr, L.send_ptr, L.send_idx, L.recv_ptr)
    return connect_halo_direct(ctx.control, dm, L)


def _local_problem(ctx, m, part, rank, world, seed=0):
    """Rank-local Poisson system with Dirichlet data on the box boundary (values from a global field)."""
    from femo_amd import engine as E
    L = build_local_mesh(m.x, m.conn, part, rank, world)
    dm = E.DeviceMesh(ctx, L.x, L.conn, n_rows=L.n_owned)
    dm.set_global(m.x.min(axis=0), m.x.max(axis=0), m.n_vert)
    if world > 1:
        _set_halo(ctx, dm, L)
    rng = np.random.default_rng(seed)
    g_global = 0.3 * rng.standard_normal(m.n_vert)
    f_global = 1.0 + rng.random(m.n_cell)
    bd = fo.boundary_vertices_box(L.x)                   # local indices, owned and ghost
    bc = E.DirichletSet(dm, bd, g_global[L.vert_global[bd]])
    nloc = len(L.x)
    A, b = E.Mat(dm), E.Vec(ctx, L.n_owned)
    E.assemble_system(dm, 0, None, E.Vec(ctx, nloc).fill(0.0), E.Vec(ctx, len(L.conn)).set(f_global[L.cell_global]),
                      bc, None, A, b)
    return L, dm, A, b, g_global, f_global


def _reference(m, seed=0):
    rng = np.random.default_rng(seed)
    g = 0.3 * rng.standard_normal(m.n_vert)
    f = 1.0 + rng.random(m.n_cell)
    bd = fo.boundary_vertices_box(m.x)
    K = fo.stiffness(m).tocsr()
    gb = np.zeros(m.n_vert)
    gb[bd] = g[bd]
    rhs = -fo.load_vector(m, f) + K @ gb          # F(0) + K[:,bc] g
    rhs[bd] = -g[bd]                              # u - g with u = 0
    return spla.spsolve(fo.eliminate_bc(K, bd).tocsc(), rhs), rhs


@pytest.mark.parametrize("halo", ["direct", "staged"])
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("pc", ["jacobi", "bpx"])
def test_partitioned_solve_with_emulated_ranks(world, pc, halo, monkeypatch):
    """`halo`: direct = device-initiated ghost refresh (stores into the neighbours' inboxes + counters, round 6);
    staged = the ncclSend/Recv-shaped neighbour exchange of the emulation (host-staged)."""
    from femo_amd import engine as E
    if halo == "staged":
        monkeypatch.setenv("FEMO_HALO_RCCL", "1")
    else:
        monkeypatch.delenv("FEMO_HALO_RCCL", raising=False)
    m = fo.unit_cube_mesh(12, 0.2)
    part = rcb_partition(m.x, world)
    x_ref, rhs_ref = _reference(m)

    def rank_fn(rank, ctx):
        L, dm, A, b, _, _ = _local_problem(ctx, m, part, rank, world)
        x = E.Vec(ctx, len(L.x))
        info = A.solve_cg(b, x, rtol=1e-14, pc=pc)
        return dict(gid=L.vert_global[:L.n_owned], x=x.get(L.n_owned), b=b.get(), its=info.iterations,
                    conv=info.converged, nbr=len(L.nbr), levels=dm.pc_info()["levels"] if pc == "bpx" else 0,
                    hd=dm.halo_direct_info())

    res = _run_ranks(world, rank_fn)
    x = np.zeros(m.n_vert)
    b = np.zeros(m.n_vert)
    for r in res:
        assert r["conv"] == 1 and r["nbr"] >= 1
        # the transport the test asked for was the one in use, and no consumer ever gave up waiting
        assert r["hd"]["enabled"] == (1 if halo == "direct" else 0) and r["hd"]["timeouts"] == 0
        assert (r["hd"]["exchanges"] > r["its"]) == (halo == "direct")
        x[r["gid"]] = r["x"]
        b[r["gid"]] = r["b"]
    assert len({r["its"] for r in res}) == 1                       # every rank stopped at the same iteration
    assert np.abs(b - rhs_ref).max() < 1e-12 * np.abs(rhs_ref).max()       # assembled right-hand side, rank by rank
    assert np.abs(x - x_ref).max() < 1e-10 * np.abs(x_ref).max()


def test_emulated_ranks_reproduce_the_single_rank_iteration_count():
    """The distributed BPX operator IS the serial one (global lattice, all-reduced accumulators): same
    iteration count as one rank, up to rounding."""
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(14, 0.2)
    counts = {}
    for world in (1, 2):
        part = rcb_partition(m.x, world)

        def rank_fn(rank, ctx):
            L, dm, A, b, _, _ = _local_problem(ctx, m, part, rank, world, seed=5)
            x = E.Vec(ctx, len(L.x))
            return A.solve_cg(b, x, rtol=1e-14, pc="bpx").iterations

        counts[world] = _run_ranks(world, rank_fn)[0]
    assert abs(counts[1] - counts[2]) <= 1, counts


def test_emulated_halo_and_dot_products():
    """SpMV with a real halo, all-reduced dot product and functional value on 2 emulated ranks."""
    from femo_amd import engine as E
    world = 2
    m = fo.unit_square_mesh(20, 0.2)
    part = rcb_partition(m.x, world)
    rng = np.random.default_rng(3)
    u = rng.standard_normal(m.n_vert)
    K = fo.stiffness(m).tocsr()

    def rank_fn(rank, ctx):
        L = build_local_mesh(m.x, m.conn, part, rank, world)
        dm = E.DeviceMesh(ctx, L.x, L.conn, n_rows=L.n_owned)
        assert _set_halo(ctx, dm, L)                       # device-initiated refresh: connected, self-test passed on both ranks
        J = E.Mat(dm)
        E.assemble_jacobian(dm, 0, None, None, None, None, J)
        ul = np.full(len(L.x), 1e30)                       # ghosts hold garbage until the exchange
        ul[:L.n_owned] = u[L.vert_global[:L.n_owned]]
        U, Y = E.Vec(ctx, len(L.x)).set(ul), E.Vec(ctx, len(L.x))
        J.mult(U, Y)
        return dict(gid=L.vert_global[:L.n_owned], y=Y.get(L.n_owned), dot=U.dot(U, L.n_owned))

    res = _run_ranks(world, rank_fn)
    y = np.zeros(m.n_vert)
    for r in res:
        y[r["gid"]] = r["y"]
        assert abs(r["dot"] - u @ u) < 1e-12 * (u @ u)              # all-reduced: every rank has the global value
    assert np.abs(y - K @ u).max() < 1e-12 * np.abs(K @ u).max()


def test_bench_cycle_on_two_emulated_ranks():
    """The SPMD path of `bench.py --gpus 2` (DistMesh through FEA / FEAModel / Simulator, BPX-CG with
    halo exchange and all-reduces) with two real ranks emulated on one GPU; state, functional and
    total derivative against the oracle's reference cycle on the whole mesh."""
    import bench as B
    from femo_amd.dist import DistMesh
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitCubeMesh
    world, n = 2, 10
    gmesh = createUnitCubeMesh(n, jitter=0.2)
    part = rcb_partition(gmesh.x, world)
    f_global = B.source_fields(gmesh, 1)[0]
    occupancy = gmesh.lattice_occupancy()

    def rank_fn(rank, ctx):
        utils_hip.set_context(ctx, thread_local=True)
        try:
            L = build_local_mesh(gmesh.x, gmesh.conn, part, rank, world)
            mesh = DistMesh(L, gmesh.n_vert, gmesh.n_cell, bbox=(gmesh.x.min(axis=0), gmesh.x.max(axis=0)))
            mesh._occupancy = occupancy
            sim, fea = B.build_problem(mesh, device=True)
            g = np.asarray(B.one_cycle(sim, fea, f_global[L.cell_global])).ravel()
            u = np.asarray(sim['u'])
            return dict(gid=L.vert_global[:L.n_owned], u=u[:L.n_owned], cells=L.cell_global[L.cell_owned],
                        g=g[L.cell_owned], J=float(np.asarray(sim['l2_functional']).ravel()[0]))
        finally:
            utils_hip.set_context(None, thread_local=True)

    res = _run_ranks(world, rank_fn)
    om = fo.unit_cube_mesh(n, jitter=0.2)
    bd = fo.boundary_vertices_box(om.x)
    ref = fo.reference_cycle(om, f_global, fo.u_target(om.x), bd, np.zeros(len(bd)))
    u = np.zeros(om.n_vert)
    g = np.zeros(om.n_cell)
    seen = np.zeros(om.n_cell, int)
    for r in res:
        u[r["gid"]] = r["u"]
        g[r["cells"]] = r["g"]
        seen[r["cells"]] += 1
        assert abs(r["J"] - ref["J"][0]) < 1e-10 * abs(ref["J"][0])          # all-reduced: global on every rank
    assert np.all(seen == 1)                                                   # every cell owned by exactly one rank
    assert np.abs(u - ref["u"]).max() < 1e-10 * np.abs(ref["u"]).max()
    assert np.abs(g - ref["grad"]).max() < 1e-10 * np.abs(ref["grad"]).max()


@pytest.mark.parametrize("world", [2, 8])
def test_sparse_and_dense_lattice_exchange_agree(world, monkeypatch):
    """Partitioned BPX exchanges only the finest-lattice nodes several ranks touch (+ the next level
    densely); FEMO_BPX_DENSE_ALLREDUCE=1 sums the whole lattice instead.  Same operator: the
    preconditioned vector agrees to rounding and so do the iteration counts."""
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(12, 0.2)
    part = rcb_partition(m.x, world)
    r_global = np.random.default_rng(11).standard_normal(m.n_vert)
    out = {}
    for mode in ("sparse", "dense"):
        if mode == "dense":
            monkeypatch.setenv("FEMO_BPX_DENSE_ALLREDUCE", "1")
        else:
            monkeypatch.delenv("FEMO_BPX_DENSE_ALLREDUCE", raising=False)

        def rank_fn(rank, ctx):
            L, dm, A, b, _, _ = _local_problem(ctx, m, part, rank, world, seed=3)
            r = E.Vec(ctx, len(L.x)).set(np.concatenate([r_global[L.vert_global[:L.n_owned]], np.zeros(len(L.x) - L.n_owned)]))
            z = A.pc_apply(r, E.Vec(ctx, len(L.x))).get(L.n_owned)
            x = E.Vec(ctx, len(L.x))
            info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
            return dict(gid=L.vert_global[:L.n_owned], z=z, its=info.iterations, x=x.get(L.n_owned))

        res = _run_ranks(world, rank_fn)
        z, x = np.zeros(m.n_vert), np.zeros(m.n_vert)
        for r in res:
            z[r["gid"]] = r["z"]
            x[r["gid"]] = r["x"]
        out[mode] = (z, x, res[0]["its"])
    monkeypatch.delenv("FEMO_BPX_DENSE_ALLREDUCE", raising=False)
    assert np.abs(out["sparse"][0] - out["dense"][0]).max() < 1e-12 * np.abs(out["dense"][0]).max()
    assert np.abs(out["sparse"][1] - out["dense"][1]).max() < 1e-11 * np.abs(out["dense"][1]).max()
    assert abs(out["sparse"][2] - out["dense"][2]) <= 1
    # and the distributed operator is the serial oracle operator
    from oracle import bpx_oracle as bo
    bd = fo.boundary_vertices_box(m.x)
    pinned = np.zeros(m.n_vert, bool)
    pinned[bd] = True
    diag = fo.eliminate_bc(fo.stiffness(m), bd).diagonal()
    ref = bo.BPX(m.x, diag, pinned).apply(r_global)
    assert np.abs(out["sparse"][0] - ref).max() < 1e-12 * np.abs(ref).max()


def test_nitsche_facets_on_partitioned_meshes():
    """Weak (Nitsche) boundary terms on 2 emulated ranks: the exterior facets are those of the WHOLE
    mesh -- the cut faces of the partition must not act as a boundary (partition_mesh hands every rank
    the global facet mask of its cells).  Newton system of the nonlinear Poisson form vs the oracle."""
    from femo_amd import engine as E
    from femo_amd.dist import partition_mesh
    from femo_amd.fea.mesh import Mesh
    world = 2
    om = fo.unit_square_mesh(16, 0.15)
    gmesh = Mesh(om.x, om.conn)
    bm = fo.boundary_facets(om)
    rng = np.random.default_rng(6)
    u = 0.3 * np.sin(3 * om.x[:, 0]) + 0.2
    f = rng.standard_normal(om.n_cell)
    uex = fo.u_exact_nl(om.x)
    J_ref = fo.nl_jacobian(om, u, bm, 10.0).tocsc()
    r_ref = fo.nl_residual(om, u, f, uex, bm, 10.0)
    x_ref = spla.spsolve(J_ref, r_ref)

    def rank_fn(rank, ctx):
        mesh = partition_mesh(gmesh, rank, world, facets=True)
        L = mesh.local
        assert np.array_equal(mesh.boundary_facet_mask(), bm[L.cell_global])
        dm = mesh.device(ctx)
        dm.set_boundary_facets(mesh.boundary_facet_mask())
        nloc = len(L.x)
        ul = np.concatenate([u[L.vert_global[:L.n_owned]], np.full(nloc - L.n_owned, 1e30)])      # ghosts refreshed by the library
        U, F, UEX = E.Vec(ctx, nloc).set(ul), E.Vec(ctx, len(L.conn)).set(f[L.cell_global]), E.Vec(ctx, nloc).set(uex[L.vert_global])
        J, b = E.Mat(dm), E.Vec(ctx, L.n_owned)
        E.assemble_system(dm, 1, [10.0], U, F, None, J, None, b, aux=UEX)
        x = E.Vec(ctx, nloc)
        info = J.solve_cg(b, x, rtol=1e-14, pc="bpx")
        return dict(gid=L.vert_global[:L.n_owned], b=b.get(), x=x.get(L.n_owned), conv=info.converged)

    res = _run_ranks(world, rank_fn)
    b, x = np.zeros(om.n_vert), np.zeros(om.n_vert)
    for r in res:
        assert r["conv"] == 1
        b[r["gid"]] = r["b"]
        x[r["gid"]] = r["x"]
    assert np.abs(b - r_ref).max() < 1e-12 * np.abs(r_ref).max()
    assert np.abs(x - x_ref).max() < 1e-9 * np.abs(x_ref).max()


def test_nonlinear_cycle_on_two_emulated_ranks():
    """BASELINE config 5's cycle (nonlinear Poisson + symmetric Nitsche, SNES, adjoint gradient) through
    FEA / FEAModel / Simulator on a 2-way partition, against the oracle's cycle on the whole mesh."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.dist import partition_mesh
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
    from femo_amd.fea.mesh import createUnitSquareMesh
    from femo_amd.fea.nonlinear_poisson import ALPHA_1, outputForm, pdeRes
    world, n = 2, 14
    gmesh = createUnitSquareMesh(n)
    gmesh.lattice_occupancy()
    gmesh.boundary_facet_mask()

    def rank_fn(rank, ctx):
        utils_hip.set_context(ctx, thread_local=True)
        try:
            mesh = partition_mesh(gmesh, rank, world)
            L = mesh.local
            fea = FEA(mesh)
            fea.REPORT = False
            Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
            f_fn, u_fn, u_ex = Function(Vf), Function(Vu), Function(Vu)
            u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]))
            fea.add_input('f', f_fn)
            fea.add_state(name='u', function=u_fn, arguments=['f'],
                          residual_form=pdeRes(u_fn, TestFunction(Vu), f_fn, u_exact=u_ex, weak_bc=True, sym=True))
            fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
            fea.PDE_SOLVER = 'SNES'
            model = FEAModel(fea=[fea])
            model.create_input('f', shape=fea.inputs_dict['f']['shape'], val=0.1)
            sim = Simulator(model, device=True)
            sim.run()
            g = np.asarray(sim.compute_totals('l2_functional', 'f')).ravel()
            u = np.asarray(sim['u'])
            return dict(gid=L.vert_global[:L.n_owned], u=u[:L.n_owned], cells=L.cell_global[L.cell_owned], g=g[L.cell_owned],
                        J=float(np.asarray(sim['l2_functional']).ravel()[0]))
        finally:
            utils_hip.set_context(None, thread_local=True)

    res = _run_ranks(world, rank_fn)
    om = fo.unit_square_mesh(n)
    ref = fo.nl_reference_cycle(om, 0.1 * np.ones(om.n_cell), fo.u_exact_nl(om.x), fo.boundary_facets(om), ALPHA_1)
    u, g = np.zeros(om.n_vert), np.zeros(om.n_cell)
    for r in res:
        u[r["gid"]] = r["u"]
        g[r["cells"]] = r["g"]
        assert abs(r["J"] - ref["J"][0]) < 1e-10 * abs(ref["J"][0])
    assert np.abs(u - ref["u"]).max() < 1e-10 * np.abs(ref["u"]).max()
    assert np.abs(g - ref["grad"]).max() < 1e-10 * np.abs(ref["grad"]).max()


@pytest.mark.parametrize("world", [2, 4])
def test_bicgstab_on_emulated_ranks(world):
    """Non-symmetric operator (unsymmetric Nitsche) on a partitioned mesh: BiCGSTAB with halo exchanges
    and all-reduced inner products against LU, for the operator AND its transpose.  The explicit transpose
    of a row-partitioned operator needs matrix entries owned by other ranks, so the transposed product is
    formed by scatter + the reverse halo add of SURVEY.md section 8(e) (round 1 refused it)."""
    from femo_amd import engine as E
    from femo_amd.dist import partition_mesh
    from femo_amd.fea.mesh import Mesh
    om = fo.unit_square_mesh(20, 0.2)
    gmesh = Mesh(om.x, om.conn)
    gmesh.lattice_occupancy()
    bm = fo.boundary_facets(om)
    rng = np.random.default_rng(9)
    u, f, uex = 0.3 * rng.standard_normal(om.n_vert), rng.standard_normal(om.n_cell), fo.u_exact_nl(om.x)
    Jo = fo.nl_jacobian(om, u, bm, 0.0, -1.0)
    rhs = rng.standard_normal(om.n_vert)
    x_ref = spla.splu(Jo.tocsc()).solve(rhs)
    xt_ref = spla.splu(Jo.T.tocsc()).solve(rhs)
    assert np.abs(x_ref - xt_ref).max() > 1e-3 * np.abs(x_ref).max()          # genuinely non-symmetric

    def rank_fn(rank, ctx):
        mesh = partition_mesh(gmesh, rank, world, facets=True)
        L = mesh.local
        dm = mesh.device(ctx)
        dm.set_boundary_facets(mesh.boundary_facet_mask())
        nloc = len(L.x)
        ul = np.concatenate([u[L.vert_global[:L.n_owned]], np.zeros(nloc - L.n_owned)])
        U, F, UEX = E.Vec(ctx, nloc).set(ul), E.Vec(ctx, len(L.conn)).set(f[L.cell_global]), E.Vec(ctx, nloc).set(uex[L.vert_global])
        J = E.Mat(dm)
        E.assemble_jacobian(dm, 1, [0.0, -1.0], U, F, None, J, aux=UEX)
        B, X, XT = E.Vec(ctx, L.n_owned).set(rhs[L.vert_global[:L.n_owned]]), E.Vec(ctx, nloc), E.Vec(ctx, nloc)
        i1 = J.solve_bicgstab(B, X, rtol=1e-13)
        i2 = J.solve_bicgstab(B, XT, transpose=True, rtol=1e-13)
        # the transposed product itself: y = J^T v against the oracle
        V, Y = E.Vec(ctx, nloc).set(np.concatenate([rhs[L.vert_global[:L.n_owned]], np.full(nloc - L.n_owned, 1e30)])), E.Vec(ctx, nloc)
        J.mult(V, Y, transpose=True)
        return dict(gid=L.vert_global[:L.n_owned], x=X.get(L.n_owned), xt=XT.get(L.n_owned), y=Y.get(L.n_owned),
                    conv=(i1.converged, i2.converged))

    res = _run_ranks(world, rank_fn)
    x, xt, y = np.zeros(om.n_vert), np.zeros(om.n_vert), np.zeros(om.n_vert)
    for r in res:
        assert r["conv"] == (1, 1)
        x[r["gid"]] = r["x"]
        xt[r["gid"]] = r["xt"]
        y[r["gid"]] = r["y"]
    assert np.abs(x - x_ref).max() < 1e-9 * np.abs(x_ref).max()
    assert np.abs(y - Jo.T @ rhs).max() < 1e-12 * np.abs(Jo.T @ rhs).max()
    assert np.abs(xt - xt_ref).max() < 1e-9 * np.abs(xt_ref).max()


def test_projection_on_two_emulated_ranks():
    """L2 projection onto CG1 (utils_dolfinx.py:549-583) on a partitioned mesh: consistent mass matrix
    (Jacobi-CG with halo + all-reduces), lumped mass, gradient magnitude of a CG1 function."""
    from femo_amd.dist import partition_mesh
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import Function, FunctionSpace, GradientMagnitude, project, setFuncArray
    from femo_amd.fea.mesh import createUnitSquareMesh
    world, n = 2, 15
    gmesh = createUnitSquareMesh(n, 0.2)
    gmesh.lattice_occupancy()
    gmesh.boundary_facet_mask()
    om = fo.unit_square_mesh(n, 0.2)
    rng = np.random.default_rng(8)
    un, wc = rng.standard_normal(om.n_vert), rng.uniform(0.5, 1.5, om.n_cell)

    def rank_fn(rank, ctx):
        utils_hip.set_context(ctx, thread_local=True)
        try:
            mesh = partition_mesh(gmesh, rank, world)
            L = mesh.local
            Vu, Vf = FunctionSpace(mesh, ('CG', 1)), FunctionSpace(mesh, ('DG', 0))
            u, w, out = Function(Vu), Function(Vf), Function(Vu)
            setFuncArray(u, un[L.vert_global])
            setFuncArray(w, wc[L.cell_global])
            got = {}
            project(w, out)
            got["dg0"] = out.vector.getArray()[:L.n_owned].copy()
            project(w, out, lump_mass=True)
            got["dg0_lumped"] = out.vector.getArray()[:L.n_owned].copy()
            project(GradientMagnitude(u), out)
            got["grad"] = out.vector.getArray()[:L.n_owned].copy()
            got["gid"] = L.vert_global[:L.n_owned]
            return got
        finally:
            utils_hip.set_context(None, thread_local=True)

    res = _run_ranks(world, rank_fn)
    refs = {"dg0": fo.project_l2(om, cell_values=wc), "dg0_lumped": fo.project_l2(om, cell_values=wc, lump_mass=True),
            "grad": fo.project_l2(om, cell_values=fo.grad_magnitude(om, un))}
    for key, ref in refs.items():
        full = np.zeros(om.n_vert)
        for r in res:
            full[r["gid"]] = r[key]
        assert np.abs(full - ref).max() < 1e-9 * np.abs(ref).max(), key


def test_distributed_bench_record_on_eight_emulated_ranks():
    """`bench.py --gpus 8` end to end with the eight ranks emulated on one GPU (bench.run_distributed_bench with a
    thread control plane instead of torch.distributed + RCCL): rank-local mesh generation under the 1 x 2 x 4
    block partition, NumPy arrays at every rank's operator boundary, and a record whose counts are consistent."""
    from argparse import Namespace
    from femo_amd.dist import ThreadControl
    from bench import run_distributed_bench
    from femo_amd.engine import Context, EmuGroup
    from femo_amd.fea import utils_hip
    world, n = 8, 48
    group = EmuGroup(world)
    shared = ThreadControl.Shared(world)
    args = Namespace(n=n, steps=2, warmup=1, pc="bpx", jitter=0.0, cpu_n=0, no_cpu_baseline=True)
    out, err = [None] * world, [None] * world

    def body(rank):
        try:
            ctx = Context(0)
            control = ThreadControl(rank, shared, group)
            control.init_comm(ctx)
            utils_hip.set_context(ctx, thread_local=True)
            try:
                out[rank] = run_distributed_bench(args, ctx, control, cpu_baseline=False)
                ctx.sync()
            finally:
                utils_hip.set_context(None, thread_local=True)
        except BaseException as e:          # noqa: BLE001
            err[rank] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for e in err:
        if e is not None:
            raise e
    assert all(o is None for o in out[1:])
    r = out[0]
    c = r["config"]
    N = (n + 1) ** 3
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and c["n_dof"] == N and c["parallelism"] == "block8"
    assert abs(r["value"] - N / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert sum(c["owned_per_rank"]) == N and max(c["owned_per_rank"]) - min(c["owned_per_rank"]) <= 3 * (n + 1) ** 2
    assert all(2 <= k <= 5 for k in c["neighbours_per_rank"])                  # 1 x 2 x 4 pencils: the y neighbour, one or two z neighbours and the diagonals the Kuhn split couples
    assert all(b > 0 for b in c["halo_bytes_sent_per_exchange_per_rank"])
    assert sum(c["halo_bytes_sent_per_exchange_per_rank"]) == sum(c["halo_bytes_received_per_exchange_per_rank"])
    # the partitioned run checks itself against the DST-exact cycle of the whole mesh (rank 0 computes, every rank compares)
    assert r["check"]["u_rel_err"] < 1e-10 and r["check"]["grad_rel_err"] < 1e-10
    assert c["linear_solves_per_step"] == 4
    its = c["cg_iterations_per_step"]
    assert len(its) == 4 and 10 < its[0] <= 40 and 10 < its[3] <= 40 and its[1] <= 2 and its[2] <= 2
    assert "host" in c["boundary"] and len(c["setup_rss_mb_per_rank"]) == 8
    # the merged loop: one all-reduce per enqueued iteration, counted inside the solver loops (a few enqueued iterations lie
    # behind the converged one in the first, unpredicted solves)
    assert 1.0 <= c["allreduce_per_cg_iteration"] <= 1.35, c["allreduce_per_cg_iteration"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] > 0 and len(rf["achieved_per_rank"]) == 8


def test_merged_pcg_one_allreduce_per_iteration_on_emulated_ranks(monkeypatch):
    """Round 4: the merged BPX-PCG keeps the restricted residual as lattice state (g -= alpha P^T q), so that p.q travels
    with the lattice sums and r.r / the lattice dot follow from reduced scalars: ONE all-reduce and one halo exchange per
    iteration.  On a mesh whose lattice has the depth the merged kernels need (n = 48: 4 levels), with 2, 4 and 8 emulated
    ranks: every rank stops in the same iteration, the count is the single-rank count (merged and classic) to within one,
    the solution is the single-rank classic solution, and the collectives are COUNTED (femo_comm_stats)."""
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(48, 0.2)
    ref = {}

    def solve(world, classic):
        if classic:
            monkeypatch.setenv("FEMO_PCG_CLASSIC", "1")
        else:
            monkeypatch.delenv("FEMO_PCG_CLASSIC", raising=False)
        part = rcb_partition(m.x, world)

        def rank_fn(rank, ctx):
            L, dm, A, b, _, _ = _local_problem(ctx, m, part, rank, world, seed=3)
            x = E.Vec(ctx, len(L.x))
            A.solve_cg(b, x, rtol=1e-11, pc="bpx")                     # first solve: lattice plan, shared-node set-up
            ctx.sync()
            ctx.comm_stats(reset=True)
            info = A.solve_cg(b, x, rtol=1e-11, pc="bpx")
            st = ctx.comm_stats()
            return dict(gid=L.vert_global[:L.n_owned], x=x.get(L.n_owned), its=info.iterations, conv=info.converged,
                        levels=dm.pc_info()["levels"], st=st, loop_ar=info.loop_allreduces)

        res = _run_ranks(world, rank_fn)
        x = np.zeros(m.n_vert)
        for r in res:
            assert r["conv"] == 1 and r["levels"] >= 4
            if world > 1 and not classic:
                # merged: one all-reduce per enqueued iteration; one halo exchange per preconditioner application (round 5: the
                # prolongation starts the exchange of the direction it produces -- the first application included)
                assert r["loop_ar"] == r["st"]["neighbor_calls"] - 1
            x[r["gid"]] = r["x"]
        assert len({r["its"] for r in res}) == 1
        return x, res[0]["its"], [r["st"] for r in res]

    x_classic, its_classic, _ = solve(1, True)
    x_merged, its_merged, _ = solve(1, False)
    assert abs(its_merged - its_classic) <= 1
    scale = np.abs(x_classic).max()
    assert np.abs(x_merged - x_classic).max() < 1e-9 * scale
    for world in (2, 4, 8):
        x, its, stats = solve(world, False)
        assert abs(its - its_merged) <= 1, (world, its, its_merged)
        assert np.abs(x - x_classic).max() < 1e-9 * scale
        for st in stats:
            # per solve: the set-up reduction of (r0.r0, b.b) and the first preconditioner application, then ONE
            # all-reduce and ONE halo exchange per enqueued iteration (the host enqueues in batches: up to a batch
            # of iterations behind the converged one is enqueued and returns at once on the device)
            enqueued = st["neighbor_calls"] - 1
            assert its <= enqueued < its + 8, (world, st, its)
            assert st["allreduce_calls"] == enqueued + 2, (world, st, its)
        _, its_c, stats_c = solve(world, True)
        assert stats_c[0]["allreduce_calls"] >= 3 * its_c        # what the classic loop issues: three per iteration


@pytest.mark.parametrize("variant", ["one_launch", "two_launches", "staged"])
def test_merged_loop_ghost_refresh_variants_agree(variant, monkeypatch):
    """Round 6: the three ways the merged BPX-PCG refreshes its ghosts on N ranks -- device-initiated with the product as ONE
    launch (ghost-column slices last, their waves wait on the counters and read the inbox), device-initiated with two launches
    and `k_halo_pull` in between (FEMO_SPMV_TWO_LAUNCHES), and the ncclSend/Recv-shaped exchange (FEMO_HALO_RCCL) -- give the
    same solve on 4 emulated ranks: same iteration count, the direct solution, one all-reduce and one refresh per iteration."""
    from femo_amd import engine as E
    monkeypatch.delenv("FEMO_HALO_RCCL", raising=False)
    monkeypatch.delenv("FEMO_SPMV_TWO_LAUNCHES", raising=False)
    if variant == "two_launches":
        monkeypatch.setenv("FEMO_SPMV_TWO_LAUNCHES", "1")
    if variant == "staged":
        monkeypatch.setenv("FEMO_HALO_RCCL", "1")
    world = 4
    m = fo.unit_cube_mesh(48, 0.2)
    part = rcb_partition(m.x, world)

    def one_rank(rank, ctx):                                          # the reference: the same solve on one rank
        L, dm, A, b, _, _ = _local_problem(ctx, m, rcb_partition(m.x, 1), 0, 1, seed=3)
        x = E.Vec(ctx, len(L.x))
        info = A.solve_cg(b, x, rtol=1e-11, pc="bpx")
        return x.get(L.n_owned)[np.argsort(L.vert_global[:L.n_owned])], info.iterations

    x_ref, its_ref = _run_ranks(1, one_rank)[0]

    def rank_fn(rank, ctx):
        L, dm, A, b, _, _ = _local_problem(ctx, m, part, rank, world, seed=3)
        x = E.Vec(ctx, len(L.x))
        A.solve_cg(b, x, rtol=1e-11, pc="bpx")
        ctx.sync()
        ctx.comm_stats(reset=True)
        info = A.solve_cg(b, x, rtol=1e-11, pc="bpx")
        return dict(gid=L.vert_global[:L.n_owned], x=x.get(L.n_owned), its=info.iterations, conv=info.converged,
                    st=ctx.comm_stats(), loop_ar=info.loop_allreduces, hd=dm.halo_direct_info())

    res = _run_ranks(world, rank_fn)
    x = np.zeros(m.n_vert)
    for r in res:
        assert r["conv"] == 1 and r["hd"]["timeouts"] == 0
        assert r["hd"]["enabled"] == (0 if variant == "staged" else 1)
        assert r["loop_ar"] == r["st"]["neighbor_calls"] - 1
        x[r["gid"]] = r["x"]
    assert len({r["its"] for r in res}) == 1 and abs(res[0]["its"] - its_ref) <= 1
    assert np.abs(x - x_ref).max() < 1e-9 * np.abs(x_ref).max()


@pytest.mark.slow
def test_eight_emulated_ranks_at_the_benchmark_size():
    """`bench.py --gpus 8`'s code path at n = 215 (10,077,696 DOFs) with the eight ranks emulated on this GPU (VERDICT round 4:
    the run was a script, scripts/run_emulated_ranks_bench.py, never executed on the driver's box): the self-check against the
    DST-exact cycle of the whole mesh, the ONE-GPU iteration counts, one all-reduce per enqueued iteration, and the size of
    that all-reduce -- round 5 exchanges all three brick-filled lattice levels sparsely (round 4: 138 k doubles per call)."""
    import json
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "run_emulated_ranks_bench.py"), "215", "8"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    c = r["config"]
    assert c["n_dof"] == 216 ** 3 and r["n_gpus"] == 8
    assert r["check"]["passed"] and r["check"]["u_rel_err"] < 1e-10 and r["check"]["grad_rel_err"] < 1e-10
    assert c["cg_iterations_per_step"] == [28, 0, 0, 28]                       # the one-GPU counts of the headline
    assert 1.0 <= c["allreduce_per_cg_iteration"] <= 1.35
    per_call = c["collectives_per_step_rank0"]["allreduce_doubles"] / c["collectives_per_step_rank0"]["allreduce_calls"]
    assert per_call < 60_000, per_call


def test_nonlinear_cycle_on_four_emulated_ranks():
    """BASELINE config 5 at its stated rank count (4): the nonlinear Poisson + symmetric Nitsche cycle (SNES, Jacobian
    reassembled every Newton step, adjoint-of-Newton gradient) on the rank-local meshes `bench.py --gpus 4` would build for
    the square (1 x 4 slabs, dist/structured.py), BPX-PCG with the merged loop -- against the oracle's cycle on the whole mesh
    and against the ONE-rank run: the same Newton step count and the same CG counts (to within one) in every solve."""
    import threading
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.dist import local_unit_mesh
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
    from femo_amd.fea.nonlinear_poisson import ALPHA_1, outputForm, pdeRes
    n = 96

    def run(world):
        def rank_fn(rank, ctx):
            utils_hip.set_context(ctx, thread_local=True)
            try:
                mesh = local_unit_mesh(n, 2, rank, world)
                L = mesh.local
                fea = FEA(mesh)
                fea.REPORT = False
                Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
                f_fn, u_fn, u_ex = Function(Vf), Function(Vu), Function(Vu)
                u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]))
                fea.add_input('f', f_fn)
                fea.add_state(name='u', function=u_fn, arguments=['f'],
                              residual_form=pdeRes(u_fn, TestFunction(Vu), f_fn, u_exact=u_ex, weak_bc=True, sym=True))
                fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
                fea.PDE_SOLVER = 'SNES'
                model = FEAModel(fea=[fea])
                model.create_input('f', shape=fea.inputs_dict['f']['shape'], val=0.1)
                sim = Simulator(model, device=True)
                me = threading.get_ident()
                sim.run()
                g = np.asarray(sim.compute_totals('l2_functional', 'f')).ravel()
                u = np.asarray(sim['u'])
                mine = [i for i in list(utils_hip.LAST_KSP_INFO) if i.get("thread") == me]
                its = [i["iterations"] for i in mine]
                loop_ar = sum(i.get("loop_allreduces", 0) for i in mine)
                return dict(gid=L.vert_global[:L.n_owned], u=u[:L.n_owned], cells=L.cell_global[L.cell_owned], g=g[L.cell_owned],
                            J=float(np.asarray(sim['l2_functional']).ravel()[0]), its=its, loop_ar=loop_ar,
                            levels=mesh.device(ctx).pc_info()["levels"])
            finally:
                utils_hip.set_context(None, thread_local=True)
        del utils_hip.LAST_KSP_INFO[:]                      # (thread idents are recycled between the two runs)
        return _run_ranks(world, rank_fn)

    one = run(1)[0]
    res = run(4)
    om = fo.unit_square_mesh(n)
    ref = fo.nl_reference_cycle(om, 0.1 * np.ones(om.n_cell), fo.u_exact_nl(om.x), fo.boundary_facets(om), ALPHA_1)
    u, g = np.zeros(om.n_vert), np.zeros(om.n_cell)
    for r in res:
        u[r["gid"]] = r["u"]
        g[r["cells"]] = r["g"]
        assert abs(r["J"] - ref["J"][0]) < 1e-10 * abs(ref["J"][0])
        assert r["levels"] >= 4
        # the partitioned run is the one-rank algorithm: as many linear solves (Newton steps + the adjoint), the same counts
        assert len(r["its"]) == len(one["its"]) and all(abs(a - b) <= 1 for a, b in zip(r["its"], one["its"])), (r["its"], one["its"])
    assert len({tuple(r["its"]) for r in res}) == 1
    # the MERGED loop runs on the partitioned 2-D lattice (round 5: three fused levels on N ranks too): one all-reduce per
    # enqueued iteration, where the classic loop issues three
    assert all(sum(r["its"]) <= r["loop_ar"] <= sum(r["its"]) + 8 * len(r["its"]) for r in res), [(r["its"], r["loop_ar"]) for r in res]
    assert np.abs(u - ref["u"]).max() < 1e-10 * np.abs(ref["u"]).max()
    assert np.abs(g - ref["grad"]).max() < 1e-10 * np.abs(ref["grad"]).max()
