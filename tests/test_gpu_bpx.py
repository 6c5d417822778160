"""The auxiliary-lattice BPX preconditioner of femo_solve_cg (csrc/bpx.hip): same answers as the
oracle's direct solve and as Jacobi-CG, mesh-independent iteration counts, loud refusal on
operators it is not meant for.  Tolerance: 1e-10 relative (BASELINE.json), both solves run to
rtol 1e-14 in the D^-1 norm."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - b).max() / np.abs(b).max()


def _poisson_system(ctx, m, seed=0):
    from femo_amd import engine as E
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    bd = fo.boundary_vertices_box(m.x)
    rng = np.random.default_rng(seed)
    g = 0.3 * rng.standard_normal(len(bd))
    bc = E.DirichletSet(dm, bd, g)
    f = 1.0 + rng.random(m.n_cell)
    u = np.zeros(m.n_vert)
    A, b = E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 0, None, E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).set(f), bc, None, A, b)
    return dm, bc, A, b


@pytest.mark.parametrize("d,n,jit", [(2, 48, 0.0), (2, 64, 0.25), (3, 12, 0.0), (3, 16, 0.25)])
def test_bpx_matches_direct_solve(ctx, d, n, jit):
    from femo_amd import engine as E
    m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    dm, bc, A, b = _poisson_system(ctx, m)
    x_ref = spla.spsolve(A.to_scipy().tocsc(), b.get())
    xj, xb = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_vert)
    ij = A.solve_cg(b, xj, rtol=1e-14, pc="jacobi")
    ib = A.solve_cg(b, xb, rtol=1e-14, pc="bpx")
    assert ij.converged == 1 and ib.converged == 1
    assert _rel(xb.get(), x_ref) < 1e-10 and _rel(xj.get(), x_ref) < 1e-10
    assert ib.iterations < ij.iterations
    assert dm.pc_info()["levels"] >= 2


def test_bpx_iterations_do_not_grow_with_the_mesh(ctx):
    from femo_amd import engine as E
    its = {}
    for n in (16, 32, 64):
        m = fo.unit_cube_mesh(n, 0.2)
        dm, bc, A, b = _poisson_system(ctx, m, seed=n)
        x = E.Vec(ctx, m.n_vert)
        info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
        assert info.converged == 1
        its[n] = info.iterations
        # residual of the returned solution against the assembled operator
        y = E.Vec(ctx, m.n_vert)
        A.mult(x, y)
        assert np.abs(y.get() - b.get()).max() <= 1e-11 * max(1.0, np.abs(b.get()).max())
    assert max(its.values()) <= 70, its
    assert its[64] <= its[16] + 15, its


def test_bpx_nonzero_guess_and_atol(ctx):
    from femo_amd import engine as E
    m = fo.unit_cube_mesh(14, 0.1)
    dm, bc, A, b = _poisson_system(ctx, m)
    x_ref = spla.spsolve(A.to_scipy().tocsc(), b.get())
    x = E.Vec(ctx, m.n_vert).set(x_ref * (1.0 + 1e-3 * np.random.default_rng(1).standard_normal(m.n_vert)))
    info = A.solve_cg(b, x, rtol=1e-14, pc="bpx", zero_guess=False)
    assert info.converged == 1 and _rel(x.get(), x_ref) < 1e-10
    # a loose absolute tolerance stops early and says so
    x2 = E.Vec(ctx, m.n_vert)
    loose = A.solve_cg(b, x2, rtol=1e-14, atol=1e-3 * info.rhs_norm, pc="bpx")
    assert loose.converged == 1 and loose.iterations < info.iterations + 40
    assert loose.residual_norm <= 1e-3 * info.rhs_norm


def test_bpx_nitsche_jacobian(ctx):
    """Weak (Nitsche) boundary conditions: the facet vertices are the pinned set."""
    from femo_amd import engine as E
    m = fo.unit_square_mesh(40, 0.15)
    bm = fo.boundary_facets(m)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    dm.set_boundary_facets(bm)
    rng = np.random.default_rng(3)
    u = 0.3 * np.sin(3 * m.x[:, 0]) + 0.2
    U, F, UEX = E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).set(rng.standard_normal(m.n_cell)), E.Vec(ctx, m.n_vert).set(fo.u_exact_nl(m.x))
    J, B = E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 1, [10.0], U, F, None, J, None, B, aux=UEX)
    x_ref = spla.spsolve(J.to_scipy().tocsc(), B.get())
    xj, xb = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_vert)
    ij = J.solve_cg(B, xj, rtol=1e-14, pc="jacobi")
    ib = J.solve_cg(B, xb, rtol=1e-14, pc="bpx")
    assert ib.converged == 1 and _rel(xb.get(), x_ref) < 1e-10
    assert ib.iterations < ij.iterations and ib.iterations <= 80


def test_bpx_refuses_other_operators(ctx):
    from femo_amd import engine as E
    from femo_amd._lib import FemoError
    m = fo.unit_square_mesh(8)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    M = E.Mat(dm)
    E.assemble_jacobian(dm, 2, None, None, None, None, M)      # mass matrix
    b, x = E.Vec(ctx, m.n_vert).fill(1.0), E.Vec(ctx, m.n_vert)
    with pytest.raises(FemoError, match="Poisson-type"):
        M.solve_cg(b, x, pc="bpx")
    assert M.solve_cg(b, x, pc="jacobi").converged == 1


@pytest.mark.parametrize("d,n", [(3, 40), (2, 160), (3, 15)])
def test_carried_x_update_is_the_old_loop(ctx, monkeypatch, d, n):
    """Round 3: x += alpha p rides in the coarse-lattice launch of the preconditioner when the fused lattice cycle runs (the
    two larger meshes; on the small one the loop is the old one either way).  Same arithmetic on the same numbers: iteration
    count and solution are those of the loop with the update in k_pcg_xr (FEMO_PCG_NO_XCARRY)."""
    from femo_amd import engine as E
    m = fo.unit_square_mesh(n, 0.1) if d == 2 else fo.unit_cube_mesh(n, 0.1)
    dm, bc, A, b = _poisson_system(ctx, m, seed=4)
    x0, x1 = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_vert)
    monkeypatch.delenv("FEMO_PCG_NO_XCARRY", raising=False)
    i0 = A.solve_cg(b, x0, rtol=1e-11, pc="bpx")
    monkeypatch.setenv("FEMO_PCG_NO_XCARRY", "1")
    i1 = A.solve_cg(b, x1, rtol=1e-11, pc="bpx")
    monkeypatch.delenv("FEMO_PCG_NO_XCARRY")
    assert i0.converged == 1 and i1.converged == 1 and i0.iterations == i1.iterations > 5
    # (not bit for bit: the brick restriction flushes its node sums with atomics, so two runs of ONE loop differ in the last bits too)
    assert _rel(x0.get(), x1.get()) < 1e-12
    x_ref = spla.spsolve(A.to_scipy().tocsc(), b.get())
    assert _rel(x0.get(), x_ref) < 1e-9


def test_bpx_allreduce_path_on_one_rank(monkeypatch):
    """FEMO_FORCE_MULTI routes the lattice and scalar all-reduces through a 1-rank communicator,
    and the self-halo mesh of test_gpu_dist exercises the overlapped SpMV inside the BPX loop."""
    from femo_amd import engine as E
    from femo_amd.engine import Context
    c = Context(0)
    c.comm_init(Context.comm_unique_id(), 0, 1)
    m = fo.unit_cube_mesh(12, 0.2)
    dm, bc, A, b = _poisson_system(c, m)
    x0, x1 = E.Vec(c, m.n_vert), E.Vec(c, m.n_vert)
    monkeypatch.delenv("FEMO_FORCE_MULTI", raising=False)
    i0 = A.solve_cg(b, x0, rtol=1e-14, pc="bpx")
    monkeypatch.setenv("FEMO_FORCE_MULTI", "1")
    i1 = A.solve_cg(b, x1, rtol=1e-14, pc="bpx")
    monkeypatch.delenv("FEMO_FORCE_MULTI")
    assert i0.converged == 1 and i1.converged == 1 and abs(i1.iterations - i0.iterations) <= 1
    assert _rel(x1.get(), x0.get()) < 1e-12


@pytest.mark.parametrize("d,n", [(2, 40), (3, 10)])
def test_operator_stack_with_bpx_matches_jacobi(ctx, d, n):
    """The FEA stack picks BPX for Poisson operators by default; state, output and total
    derivative agree with the Jacobi-CG run of the same cycle to solver tolerance."""
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    from tests.test_gpu_operators import make_sim
    assert utils_hip.KSP_OPTIONS["pc"] == "bpx"
    utils_hip.set_context(ctx)
    out = {}
    for pc in ("bpx", "jacobi"):
        mesh = createUnitSquareMesh(n) if d == 2 else createUnitCubeMesh(n)
        old = utils_hip.KSP_OPTIONS["pc"]
        utils_hip.KSP_OPTIONS["pc"] = pc
        try:
            sim, fea, f_ex, u_ex = make_sim(mesh, True)
            rng = np.random.default_rng(11)
            sim['f'] = 0.086 * (1.0 + 0.3 * rng.uniform(-1, 1, mesh.n_cell))
            del utils_hip.LAST_KSP_INFO[:]
            sim.run()
            g = np.asarray(sim.compute_totals('l2_functional', 'f')).ravel().copy()
            out[pc] = (np.asarray(sim['u']).copy(), float(np.asarray(sim['l2_functional_output_model.l2_functional']).ravel()[0]), g,
                       [i["iterations"] for i in utils_hip.LAST_KSP_INFO])
        finally:
            utils_hip.KSP_OPTIONS["pc"] = old
    ub, jb, gb, itb = out["bpx"]
    uj, jj, gj, itj = out["jacobi"]
    assert _rel(ub, uj) < 1e-10 and abs(jb - jj) <= 1e-10 * abs(jj) and _rel(gb, gj) < 1e-9
    assert max(itb) < max(itj)


@pytest.mark.parametrize("d,n,jit", [(2, 37, 0.2), (2, 64, 0.0), (3, 9, 0.25), (3, 20, 0.2), (3, 32, 0.0)])
def test_pc_apply_matches_the_oracle_operator(ctx, d, n, jit):
    """femo_mat_pc_apply = the operator oracle/bpx_oracle.py writes down (lattice choice, packed
    coordinates, keep rule, level weights, nested transfers), to rounding error; and it is symmetric."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    dm, bc, A, b = _poisson_system(ctx, m)
    pinned = np.zeros(m.n_vert, bool)
    pinned[fo.boundary_vertices_box(m.x)] = True
    diag = A.to_scipy().diagonal()
    M = bo.BPX(m.x, diag, pinned)
    rng = np.random.default_rng(4)
    R, Z = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_vert)
    zs = []
    for k in range(2):
        r = rng.standard_normal(m.n_vert)
        A.pc_apply(R.set(r), Z)
        z = Z.get()
        ref = M.apply(r)
        assert np.abs(z - ref).max() < 1e-12 * np.abs(ref).max()
        zs.append((r, z))
    info = dm.pc_info()
    assert info["levels"] == M.levels and info["finest_nodes"] == int(np.prod(M.bins[-1] + 1))
    (r0, z0), (r1, z1) = zs
    assert abs(r1 @ z0 - r0 @ z1) < 1e-12 * abs(r1 @ z0)          # <r1, M^-1 r0> = <r0, M^-1 r1>


def test_pc_apply_with_nitsche_pinning(ctx):
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    m = fo.unit_cube_mesh(10, 0.15)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    dm.set_boundary_facets(fo.boundary_facets(m))
    u = 0.3 * np.sin(3 * m.x[:, 0]) + 0.2
    U, F, UEX = E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).fill(0.5), E.Vec(ctx, m.n_vert).set(fo.u_exact_nl(m.x))
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 1, [10.0], U, F, None, J, aux=UEX)
    pinned = np.zeros(m.n_vert, bool)
    pinned[fo.boundary_vertices_box(m.x)] = True
    M = bo.BPX(m.x, J.to_scipy().diagonal(), pinned)
    r = np.random.default_rng(2).standard_normal(m.n_vert)
    z = J.pc_apply(E.Vec(ctx, m.n_vert).set(r), E.Vec(ctx, m.n_vert)).get()
    ref = M.apply(r)
    assert np.abs(z - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("d,n", [(2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3)])
def test_bpx_on_tiny_meshes(ctx, d, n):
    """Sub-wave meshes: one or two lattice levels, bricks with a handful of vertices, nearly (or
    entirely) pinned vertex sets.  Same answer as the direct solve."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    m = fo.unit_square_mesh(n, 0.1) if d == 2 else fo.unit_cube_mesh(n, 0.1)
    dm, bc, A, b = _poisson_system(ctx, m)
    x_ref = spla.spsolve(A.to_scipy().tocsc(), b.get())
    x = E.Vec(ctx, m.n_vert)
    info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
    assert info.converged == 1 and _rel(x.get(), x_ref) < 1e-12
    pinned = np.zeros(m.n_vert, bool)
    pinned[fo.boundary_vertices_box(m.x)] = True
    M = bo.BPX(m.x, A.to_scipy().diagonal(), pinned)
    r = np.random.default_rng(0).standard_normal(m.n_vert)
    z = A.pc_apply(E.Vec(ctx, m.n_vert).set(r), E.Vec(ctx, m.n_vert)).get()
    assert np.abs(z - M.apply(r)).max() < 1e-12 * np.abs(r).max()


def test_bpx_without_any_pinned_vertex(ctx):
    """Pure Neumann stiffness + mass shift (SPD, no Dirichlet set): no mask, every lattice node kept."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    m = fo.unit_cube_mesh(9, 0.2)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    # NL Poisson Jacobian without facets: K + 3 u^2 mass term, u = 1 -> SPD without any boundary condition
    U, F = E.Vec(ctx, m.n_vert).fill(1.0), E.Vec(ctx, m.n_cell).fill(0.0)
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 1, [0.0], U, F, None, J)
    Js = J.to_scipy()
    b = np.random.default_rng(3).standard_normal(m.n_vert)
    x_ref = spla.spsolve(Js.tocsc(), b)
    x = E.Vec(ctx, m.n_vert)
    info = J.solve_cg(E.Vec(ctx, m.n_vert).set(b), x, rtol=1e-14, pc="bpx")
    assert info.converged == 1 and _rel(x.get(), x_ref) < 1e-11
    M = bo.BPX(m.x, Js.diagonal(), None)
    z = J.pc_apply(E.Vec(ctx, m.n_vert).set(b), E.Vec(ctx, m.n_vert)).get()
    assert np.abs(z - M.apply(b)).max() < 1e-12 * np.abs(M.apply(b)).max()


def _carved_mesh(d, n, jit):
    """A non-convex domain: the unit square / cube with one quadrant / octant removed, vertices
    renumbered compactly; returns the mesh and its boundary vertices (from the facet count)."""
    m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    xc = m.x[m.conn].mean(axis=1)
    keep = ~np.all(xc > 0.5, axis=1)
    conn = m.conn[keep]
    used = np.unique(conn)
    new = -np.ones(m.n_vert, np.int64)
    new[used] = np.arange(len(used))
    cm = fo.OMesh(d, m.x[used], new[conn].astype(np.int32))
    bm = fo.boundary_facets(cm)
    bv = set()
    for k in range(d + 1):
        cells = np.nonzero(bm & (1 << k))[0]
        bv.update(np.delete(cm.conn[cells], k, axis=1).ravel().tolist())
    return cm, np.array(sorted(bv), np.int32)


@pytest.mark.parametrize("d,n", [(2, 40), (3, 14)])
def test_bpx_on_a_non_convex_domain(ctx, d, n):
    """L-shaped domain: a quarter of the lattice lies outside the mesh (nodes without vertex mass
    are dropped) and the re-entrant boundary cuts through lattice bins.  Operator = oracle operator,
    solution = direct solve, far fewer iterations than Jacobi."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    cm, bd = _carved_mesh(d, n, 0.15)
    dm = E.DeviceMesh(ctx, cm.x, cm.conn)
    bc = E.DirichletSet(dm, bd, 0.0)
    rng = np.random.default_rng(8)
    A, b = E.Mat(dm), E.Vec(ctx, cm.n_vert)
    E.assemble_system(dm, 0, None, E.Vec(ctx, cm.n_vert).fill(0.0), E.Vec(ctx, cm.n_cell).set(1.0 + rng.random(cm.n_cell)),
                      bc, None, A, b)
    As = A.to_scipy()
    x_ref = spla.spsolve(As.tocsc(), b.get())
    xj, xb = E.Vec(ctx, cm.n_vert), E.Vec(ctx, cm.n_vert)
    ij = A.solve_cg(b, xj, rtol=1e-14, pc="jacobi")
    ib = A.solve_cg(b, xb, rtol=1e-14, pc="bpx")
    assert ib.converged == 1 and _rel(xb.get(), x_ref) < 1e-10
    assert ib.iterations < ij.iterations and ib.iterations <= 70
    pinned = np.zeros(cm.n_vert, bool)
    pinned[bd] = True
    M = bo.BPX(cm.x, As.diagonal(), pinned)
    r = rng.standard_normal(cm.n_vert)
    z = A.pc_apply(E.Vec(ctx, cm.n_vert).set(r), E.Vec(ctx, cm.n_vert)).get()
    assert np.abs(z - M.apply(r)).max() < 1e-12 * np.abs(M.apply(r)).max()


def test_solver_layer_keeps_jacobi_on_strongly_graded_meshes(ctx, monkeypatch):
    """BPX has no levels between its finest lattice and the local mesh size: on a mesh graded like
    x -> x^2.5 it needs more iterations than Jacobi, so KSP falls back (Mesh.lattice_occupancy)."""
    from femo_amd import engine as E
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import Mesh, createUnitCubeMesh
    utils_hip.set_context(ctx)
    base = createUnitCubeMesh(14)
    seen = []
    real = E.Mat.solve_cg

    def spy(self, b, x, **kw):
        seen.append(kw.get("pc"))
        return real(self, b, x, **kw)

    monkeypatch.setattr(E.Mat, "solve_cg", spy)
    for power, expect in ((1.0, "bpx"), (2.5, "jacobi")):
        mesh = Mesh(base.x ** power, base.conn)
        om = fo.OMesh(3, mesh.x, mesh.conn)
        dm = mesh.device(ctx)
        bd = fo.boundary_vertices_box(mesh.x)
        A = utils_hip.SparseMatrix(mesh, symmetric=True)
        A.pde_kind = 0
        b = E.Vec(ctx, mesh.n_vert)
        E.assemble_system(dm, 0, None, E.Vec(ctx, mesh.n_vert).fill(0.0), E.Vec(ctx, mesh.n_cell).fill(1.0),
                          E.DirichletSet(dm, bd, 0.0), None, A.mat, b)
        x = E.Vec(ctx, mesh.n_vert)
        utils_hip.KSP(A).solve(b, x)
        assert seen[-1] == expect
        x_ref = spla.spsolve(fo.eliminate_bc(fo.stiffness(om), bd).tocsc(), b.get())
        assert _rel(x.get(), x_ref) < 1e-10


@pytest.mark.parametrize("fused", ["0", "1", "2"])
def test_pc_apply_does_not_depend_on_the_fusion_depth(ctx, monkeypatch, fused):
    """FEMO_BPX_FUSED: how many coarser lattices the brick kernel restricts to itself (0 on
    partitioned meshes, where every fused level would enlarge the all-reduce).  Same operator."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    monkeypatch.setenv("FEMO_BPX_FUSED", fused)
    for d, n in ((3, 18), (2, 50)):
        m = fo.unit_cube_mesh(n, 0.2) if d == 3 else fo.unit_square_mesh(n, 0.2)
        dm, bc, A, b = _poisson_system(ctx, m)
        pinned = np.zeros(m.n_vert, bool)
        pinned[fo.boundary_vertices_box(m.x)] = True
        M = bo.BPX(m.x, A.to_scipy().diagonal(), pinned)
        r = np.random.default_rng(7).standard_normal(m.n_vert)
        z = A.pc_apply(E.Vec(ctx, m.n_vert).set(r), E.Vec(ctx, m.n_vert)).get()
        assert np.abs(z - M.apply(r)).max() < 1e-12 * np.abs(M.apply(r)).max()
        x = E.Vec(ctx, m.n_vert)
        info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
        assert info.converged == 1 and info.iterations <= 60


@pytest.mark.parametrize("seed", range(10))
def test_pc_apply_sweep_over_shapes(ctx, seed):
    """Randomised sweep: dimension, resolution, jitter, anisotropic scaling and translation of the
    domain (negative coordinates, extents that are not multiples of one another).  The HIP operator
    must equal the oracle operator and PCG must reach the direct solution."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    rng = np.random.default_rng(1000 + seed)
    d = 2 if seed % 2 == 0 else 3
    n = int(rng.integers(6, 40)) if d == 2 else int(rng.integers(4, 15))
    m0 = fo.unit_square_mesh(n, float(rng.uniform(0, 0.3))) if d == 2 else fo.unit_cube_mesh(n, float(rng.uniform(0, 0.3)))
    scale = rng.uniform(0.3, 3.0, size=d)
    shift = rng.uniform(-5.0, 5.0, size=d)
    m = fo.OMesh(d, m0.x * scale + shift, m0.conn)
    bd = fo.boundary_vertices_box(m0.x)                       # topology of the unit box
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    bc = E.DirichletSet(dm, bd, 0.3 * rng.standard_normal(len(bd)))
    A, b = E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 0, None, E.Vec(ctx, m.n_vert).fill(0.0), E.Vec(ctx, m.n_cell).set(1.0 + rng.random(m.n_cell)),
                      bc, None, A, b)
    As = A.to_scipy()
    pinned = np.zeros(m.n_vert, bool)
    pinned[bd] = True
    M = bo.BPX(m.x, As.diagonal(), pinned)
    r = rng.standard_normal(m.n_vert)
    z = A.pc_apply(E.Vec(ctx, m.n_vert).set(r), E.Vec(ctx, m.n_vert)).get()
    ref = M.apply(r)
    assert np.abs(z - ref).max() < 1e-11 * np.abs(ref).max()
    info_bins = dm.pc_info()
    assert info_bins["levels"] == M.levels and info_bins["finest_nodes"] == int(np.prod(M.bins[-1] + 1))
    x = E.Vec(ctx, m.n_vert)
    info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
    x_ref = spla.spsolve(As.tocsc(), b.get())
    assert info.converged == 1 and _rel(x.get(), x_ref) < 1e-10
