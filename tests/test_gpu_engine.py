"""GPU parity of the C-ABI kernels against the CPU oracle (same seeded inputs)."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu

RTOL = 1e-12  # fp64 round-off level parity between two summation orders


def _mesh(d, n, jitter):
    return fo.unit_square_mesh(n, jitter) if d == 2 else fo.unit_cube_mesh(n, jitter)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# all but the smallest have slices with regular column offsets (delta-compressed SpMV path)
CASES = [(2, 5, 0.0), (2, 33, 0.2), (3, 4, 0.0), (3, 13, 0.2), (2, 64, 0.0), (2, 200, 0.2), (3, 70, 0.0)]


@pytest.mark.parametrize("d,n,jit", CASES)
def test_assembly_parity(ctx, d, n, jit):
    from femo_amd import engine as E
    m = _mesh(d, n, jit)
    rng = np.random.default_rng(7)
    u = rng.standard_normal(m.n_vert)
    f = rng.standard_normal(m.n_cell)
    ud = rng.standard_normal(m.n_vert)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    # structured numbering: every full slice away from the first / last rows is regular (boundary rows are completed with
    # structural zeros, topology.cpp); meshes below one full slice have none
    assert (dm.info['regular_slices'] > 0) == (m.n_vert >= 256)
    K = fo.stiffness(m)
    rowptr, col = dm.pattern_csr()
    assert np.array_equal(rowptr, K.indptr) and np.array_equal(col, K.indices)
    U, F, UD = E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).set(f), E.Vec(ctx, m.n_vert).set(ud)
    R = E.Vec(ctx, m.n_vert)
    E.assemble_residual(dm, 0, None, U, F, R)
    assert _rel(R.get(), fo.residual(m, u, f)) < RTOL
    # dR/du without BCs
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, U, F, None, J)
    Kg = J.to_scipy()
    assert np.array_equal(Kg.indices, K.indices)
    assert _rel(Kg.data, K.data) < RTOL
    assert abs(Kg - Kg.T).max() == 0.0  # bitwise symmetric by construction
    # with BCs
    bdofs = fo.boundary_vertices_box(m.x)
    bc = E.DirichletSet(dm, bdofs, 0.0)
    A = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, U, F, bc, A)
    Ao = fo.eliminate_bc(K, bdofs)
    Ag = A.to_scipy()
    assert _rel(Ag.data, Ao.data) < RTOL
    # SpMV
    Y = E.Vec(ctx, m.n_vert)
    J.mult(U, Y)
    assert _rel(Y.get(), K @ u) < RTOL
    J.mult(U, Y, transpose=True)
    assert _rel(Y.get(), K.T @ u) < RTOL
    # dR/df
    d1 = d + 1
    V = E.Vec(ctx, m.n_cell * d1)
    E.assemble_dRdf(dm, 0, None, U, F, V)
    Do = fo.dRdf(m)
    vals = V.get().reshape(m.n_cell, d1)
    import scipy.sparse as sp
    Dg = sp.coo_matrix((vals.ravel(), (m.conn.ravel(), np.repeat(np.arange(m.n_cell), d1))),
                       shape=(m.n_vert, m.n_cell)).tocsr()
    assert abs(Dg - Do).max() <= RTOL * abs(Do).max()
    YC = E.Vec(ctx, m.n_cell)
    E.dRdf_apply(dm, V, U, YC, transpose=True)
    assert _rel(YC.get(), Do.T @ u) < RTOL
    E.dRdf_apply(dm, V, F, Y, transpose=False)
    assert _rel(Y.get(), Do @ f) < RTOL
    # functional
    alpha = 1e-3
    Jv = E.functional_value(dm, 0, [alpha], U, F, UD)
    assert abs(Jv - fo.functional(m, u, f, ud, alpha)) < RTOL * abs(Jv)
    G = E.Vec(ctx, m.n_vert)
    E.functional_grad_u(dm, 0, [alpha], U, F, UD, G)
    assert _rel(G.get(), fo.functional_du(m, u, ud)) < RTOL
    GF = E.Vec(ctx, m.n_cell)
    E.functional_grad_f(dm, 0, [alpha], U, F, UD, GF)
    assert _rel(GF.get(), fo.functional_df(m, f, alpha)) < RTOL
    # Newton right-hand side with lifting
    g = rng.standard_normal(len(bdofs))
    bc2 = E.DirichletSet(dm, bdofs, g)
    B = E.Vec(ctx, m.n_vert)
    E.newton_rhs(J, R, U, bc2, B)
    b_ref = fo.newton_rhs(K, fo.residual(m, u, f), u, bdofs, g)
    assert _rel(B.get(), b_ref) < RTOL
    # fused pass: dR/du, A and the Newton right-hand side in one launch
    J2, A2, B2 = E.Mat(dm), E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 0, None, U, F, bc2, J2, A2, B2)
    assert np.array_equal(J2.to_scipy().data, Kg.data)
    assert np.array_equal(A2.to_scipy().data, Ag.data)
    assert _rel(B2.get(), b_ref) < RTOL
    B3 = E.Vec(ctx, m.n_vert).set(fo.residual(m, u, f))
    E.bc_apply_rhs(bc2, U, B3)
    ref3 = fo.residual(m, u, f)
    ref3[bdofs] = u[bdofs] - g
    assert _rel(B3.get(), ref3) < RTOL


@pytest.mark.parametrize("d,n,jit", [(2, 32, 0.0), (3, 12, 0.2), (3, 24, 0.0)])
def test_cg_parity(ctx, d, n, jit):
    from femo_amd import engine as E
    import scipy.sparse.linalg as spla
    m = _mesh(d, n, jit)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    bdofs = fo.boundary_vertices_box(m.x)
    bc = E.DirichletSet(dm, bdofs, 0.0)
    A = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, bc, A)
    Ao = fo.eliminate_bc(fo.stiffness(m), bdofs)
    rng = np.random.default_rng(3)
    b = rng.standard_normal(m.n_vert)
    B, X = E.Vec(ctx, m.n_vert).set(b), E.Vec(ctx, m.n_vert)
    info = A.solve_cg(B, X, rtol=1e-13)
    xo = spla.splu(Ao.tocsc()).solve(b)
    assert info.converged == 1
    assert _rel(X.get(), xo) < 1e-10
    _, it_o, _ = fo.pcg_jacobi(Ao, b, rtol=1e-13)
    assert abs(info.iterations - it_o) <= max(2, it_o // 50)
    # transposed solve on the symmetric operator and warm start from the solution
    info2 = A.solve_cg(B, X, transpose=True, rtol=1e-13, zero_guess=False)
    assert info2.converged == 1 and info2.iterations <= 2
    # zero right-hand side converges immediately
    Z = E.Vec(ctx, m.n_vert)
    info3 = A.solve_cg(Z, X, rtol=1e-13)
    assert info3.iterations == 0 and np.all(X.get() == 0.0)


def test_spmv_column_encodings(ctx):
    """The three column encodings of the SELL SpMV on one mesh: per-slice deltas (structured numbering, boundary rows
    completed with structural zeros), 16-bit deltas (Morton numbering: every column within +-32767 of its row), 32-bit
    indices (random numbering) -- same product, and the Jacobian stays bitwise symmetric in each."""
    from femo_amd import engine as E
    from femo_amd.fea.mesh import createUnitCubeMesh
    base = createUnitCubeMesh(44, 0.2)
    rng = np.random.default_rng(3)
    seen = {}
    for name, mesh in (("structured", base), ("random", base.permuted(seed=5)), ("morton", base.permuted(seed=5).reordered())):
        dm = E.DeviceMesh(ctx, mesh.x, mesh.conn)
        ns = dm.info["n_slices"]
        seen[name] = (dm.info["regular_slices"] / ns, dm.info["short_slices"] / ns)
        om = fo.OMesh(3, mesh.x, mesh.conn)
        K = fo.stiffness(om)
        J = E.Mat(dm)
        U = E.Vec(ctx, mesh.n_vert).set(rng.standard_normal(mesh.n_vert))
        E.assemble_jacobian(dm, 0, None, U, E.Vec(ctx, mesh.n_cell), None, J)
        Kg = J.to_scipy()
        assert np.array_equal(Kg.indices, K.indices) and _rel(Kg.data, K.data) < RTOL and abs(Kg - Kg.T).max() == 0.0
        u = np.array(U.get())
        Y = E.Vec(ctx, mesh.n_vert)
        J.mult(U, Y)
        assert _rel(Y.get(), K @ u) < RTOL
        J.mult(U, Y, transpose=True)
        assert _rel(Y.get(), K.T @ u) < RTOL
    assert seen["structured"][0] > 0.97 and seen["random"] == (0.0, 0.0) and seen["morton"][0] == 0.0 and seen["morton"][1] > 0.7


@pytest.mark.parametrize("d,n,jit", [(2, 33, 0.2), (3, 13, 0.2)])
def test_compact_dRdf(ctx, d, n, jit):
    """dR/df of the Poisson forms as one value per cell (femo_assemble_dRdf_cell / femo_dRdf_cell_apply): the same matrix
    and the same products as the (d+1)-values-per-cell form and as the oracle (state_model.py:136-146, 176-200)."""
    from femo_amd import engine as E
    m = _mesh(d, n, jit)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    D = fo.dRdf(m)
    rng = np.random.default_rng(11)
    cv = E.assemble_dRdf_cell(dm, 0, None, E.Vec(ctx, m.n_cell))
    full = E.assemble_dRdf(dm, 0, None, None, None, E.Vec(ctx, m.n_cell * (d + 1)))
    assert np.array_equal(np.repeat(np.asarray(cv.get()), d + 1), np.asarray(full.get()))
    lam, df = rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell)
    L, DF = E.Vec(ctx, m.n_vert).set(lam), E.Vec(ctx, m.n_cell).set(df)
    yT = E.dRdf_cell_apply(dm, cv, L, E.Vec(ctx, m.n_cell), transpose=True)
    assert _rel(yT.get(), D.T @ lam) < RTOL
    yN = E.dRdf_cell_apply(dm, cv, DF, E.Vec(ctx, m.n_vert), transpose=False)
    assert _rel(yN.get(), D @ df) < RTOL
    acc = E.Vec(ctx, m.n_vert).set(lam)
    E.dRdf_cell_apply(dm, cv, DF, acc, transpose=False, accumulate=True)
    assert _rel(acc.get(), lam + D @ df) < RTOL
    accT = E.Vec(ctx, m.n_cell).set(df)
    E.dRdf_cell_apply(dm, cv, L, accT, transpose=True, accumulate=True)
    assert _rel(accT.get(), df + D.T @ lam) < RTOL
    with pytest.raises(E.FemoError):
        E.assemble_dRdf_cell(dm, 3, None, E.Vec(ctx, m.n_cell))          # the beam's dR/dt is not uniform per cell
