"""Pins the CPU oracle (NumPy and C ports) against closed-form known answers and the
committed golden vectors.  Parity against FEniCSx itself is UNPINNED (not installable
here; the reference ships no fixtures) -- see oracle/femo_oracle.py."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import c_port
from oracle import femo_oracle as fo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ref_simplex(d):
    x = np.vstack([np.zeros(d), np.eye(d)])
    return fo.OMesh(d, x, np.arange(d + 1, dtype=np.int32)[None, :])


def test_element_kats():
    k = np.load(os.path.join(GOLD, "element_kats.npz"))
    for d, tag in [(2, "tri"), (3, "tet")]:
        m = _ref_simplex(d)
        assert np.allclose(fo.stiffness(m).toarray(), k[f"k_{tag}"], rtol=0, atol=1e-15)
        assert np.allclose(-fo.dRdf(m).toarray().ravel(), k[f"load_{tag}"], rtol=0, atol=1e-16)
        # functional Hessian in u is the P1 mass matrix
        M = np.array([fo.functional_du(m, e, np.zeros(d + 1)) for e in np.eye(d + 1)])
        assert np.allclose(M, k[f"m_{tag}"], rtol=0, atol=1e-16)
        # residual of a linear field against a constant source
        u = m.x @ np.arange(1, d + 1)
        assert np.allclose(fo.residual(m, u, np.array([2.0])), k[f"k_{tag}"] @ u - 2.0 * k[f"load_{tag}"], atol=1e-15)


@pytest.mark.parametrize("d,n", [(2, 16), (3, 8)])
def test_structured_mesh_counts_and_stencil(d, n):
    m = fo.unit_square_mesh(n) if d == 2 else fo.unit_cube_mesh(n)
    K = fo.stiffness(m)
    nv = (n + 1) ** d
    nnz = nv + 2 * (2 * n * (n + 1) + n * n) if d == 2 else nv + 2 * (3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n ** 3)
    assert m.n_vert == nv and m.n_cell == (2 * n * n if d == 2 else 6 * n ** 3) and K.nnz == nnz
    assert abs(K @ np.ones(nv)).max() < 1e-13 and abs(K - K.T).max() < 1e-15
    # interior rows reproduce the 5-point stencil (2-D) / h x 7-point stencil (3-D)
    i = np.ravel_multi_index((n // 2,) * d, (n + 1,) * d)
    row = K[i].toarray().ravel()
    h = 1.0 / n
    assert np.isclose(row[i], 4.0 if d == 2 else 6.0 * h)
    assert np.isclose(np.sort(row)[:2 * d], -1.0 if d == 2 else -h).all()
    assert np.isclose(fo.cell_geometry(m)[0].sum(), 1.0)


@pytest.mark.parametrize("d,n", [(2, 24), (3, 10)])
def test_newton_matches_dst_exact_and_pcg(d, n):
    m = fo.unit_square_mesh(n) if d == 2 else fo.unit_cube_mesh(n)
    bd = fo.boundary_vertices_box(m.x)
    f = fo.f_star(fo.centroids(m))
    u, info = fo.newton_solve(m, f, np.ones(m.n_vert), bd, np.zeros(len(bd)))
    assert info.newton_its == 3                       # utils_dolfinx.py:419-449: always 3
    assert info.residual_norms[1] < 1e-12 * info.residual_norms[0]
    b = fo.load_vector(m, f)
    b[bd] = 0.0
    assert np.abs(fo.dst_solve(m, b) - u).max() < 1e-13 * np.abs(u).max()
    A = fo.eliminate_bc(fo.stiffness(m), bd)
    x, it, _ = fo.pcg_jacobi(A, b, rtol=1e-14)
    assert np.abs(x - u).max() < 1e-11 * np.abs(u).max() and it > 0
    # P1 interpolation error of the smooth target is O(h^2)
    assert np.abs(u - fo.u_target(m.x)).max() < 2.0 * (1.0 / n) ** 2


def test_adjoint_gradient_vs_finite_differences():
    m = fo.unit_square_mesh(8, jitter=0.2)
    bd = fo.boundary_vertices_box(m.x)
    g0 = np.zeros(len(bd))
    rng = np.random.default_rng(0)
    f = 0.086 * (1 + 0.3 * rng.uniform(-1, 1, m.n_cell))
    ud = fo.u_target(m.x)
    alpha = 1e-3

    def J(ff):
        u, _ = fo.newton_solve(m, ff, np.zeros(m.n_vert), bd, g0)
        return fo.functional(m, u, ff, ud, alpha)

    u, _ = fo.newton_solve(m, f, np.zeros(m.n_vert), bd, g0)
    g_exact, _ = fo.total_gradient(m, f, u, ud, bd, alpha, consistent_bc=True)
    g_ref, _ = fo.total_gradient(m, f, u, ud, bd, alpha, consistent_bc=False)
    for _ in range(3):
        d = rng.standard_normal(m.n_cell)
        fd = (J(f + 1e-5 * d) - J(f - 1e-5 * d)) / 2e-5
        assert abs(fd - g_exact @ d) < 1e-7 * abs(fd)
    # the reference's un-eliminated Dirichlet rows (state_model.py:132-146) leak lam on the boundary
    assert 1e-4 < np.abs(g_ref - g_exact).max() / np.abs(g_exact).max() < 0.2
    # fwd-mode quirk of the reference (fea_dolfinx.py:192-206) returns zeros
    A = fo.eliminate_bc(fo.stiffness(m), bd)
    assert np.all(fo.solve_linear_fwd_reference(A, np.ones(m.n_vert)) == 0.0)
    assert np.allclose(A @ fo.solve_linear_fwd_intended(A, np.ones(m.n_vert)), 1.0)


def test_dirichlet_algebra_with_inhomogeneous_values():
    m = fo.unit_cube_mesh(4, jitter=0.2)
    bd = fo.boundary_vertices_box(m.x)
    g = np.sin(3 * m.x[bd, 0]) + m.x[bd, 1]
    f = np.ones(m.n_cell)
    u, _ = fo.newton_solve(m, f, np.full(m.n_vert, 0.1), bd, g, initialize=True)
    assert np.abs(u[bd] - g).max() < 1e-14
    interior = np.setdiff1d(np.arange(m.n_vert), bd)
    assert np.abs(fo.residual(m, u, f)[interior]).max() < 1e-13
    A = fo.eliminate_bc(fo.stiffness(m), bd).toarray()
    assert np.all(A[bd][:, interior] == 0) and np.all(A[interior][:, bd] == 0) and np.all(np.diag(A)[bd] == 1)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "poisson_*.npz"))))
def test_golden_vectors(path):
    g = np.load(path)
    d = g["x"].shape[1]
    m = fo.OMesh(d, g["x"], g["conn"])
    bd = g["bc_dofs"]
    lin = fo.linearize(m, bd)
    assert np.array_equal(lin.dRdu.indptr, g["dRdu_indptr"]) and np.array_equal(lin.dRdu.indices, g["dRdu_indices"])
    for name, got in [("dRdu_data", lin.dRdu.data), ("A_data", lin.A.data), ("dRdf_data", lin.dRdf.data)]:
        assert np.abs(got - g[name]).max() <= 1e-14 * np.abs(g[name]).max()
    ref = fo.reference_cycle(m, g["f"], g["u_d"], bd, np.zeros(len(bd)))
    for k in ("u", "J", "grad", "lam"):
        assert np.abs(ref[k] - g[k]).max() <= 1e-12 * np.abs(g[k]).max()
    assert np.abs(fo.residual(m, g["u_rand"], g["f"]) - g["residual_u_rand"]).max() < 1e-13
    # the C port reproduces the same vectors
    rp, col = c_port.pattern(d, m.n_vert, m.conn)
    assert np.array_equal(rp, g["dRdu_indptr"]) and np.array_equal(col, g["dRdu_indices"])
    assert np.abs(c_port.stiffness(d, m.x, m.conn, rp, col) - g["dRdu_data"]).max() < 1e-14 * np.abs(g["dRdu_data"]).max()
    out = c_port.poisson_cycle(d, m.x, m.conn, g["f"], g["u_d"], bd, fo.ALPHA_POISSON)
    for k in ("u", "grad", "lam"):
        assert np.abs(out[k] - g[k]).max() <= 1e-11 * np.abs(g[k]).max()
    assert abs(out["J"] - g["J"][0]) <= 1e-12 * abs(g["J"][0])


def test_c_port_pieces():
    m = fo.unit_cube_mesh(5, jitter=0.2)
    rng = np.random.default_rng(2)
    u, f = rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell)
    assert np.abs(c_port.residual(3, m.x, m.conn, u, f) - fo.residual(m, u, f)).max() < 1e-13
    K = fo.stiffness(m)
    rp, col = c_port.pattern(3, m.n_vert, m.conn)
    val = c_port.stiffness(3, m.x, m.conn, rp, col)
    bd = fo.boundary_vertices_box(m.x)
    isbc = np.zeros(m.n_vert, np.uint8)
    isbc[bd] = 1
    assert np.abs(c_port.eliminate_bc(rp, col, val, isbc) - fo.eliminate_bc(K, bd).data).max() < 1e-14
    assert np.abs(c_port.spmv(rp, col, val, u) - K @ u).max() < 1e-13
    A = sp.csr_matrix((c_port.eliminate_bc(rp, col, val, isbc), col, rp))
    b = rng.standard_normal(m.n_vert)
    x, it, res = c_port.pcg(rp, col, A.data, b, rtol=1e-13)
    xo, it_o, _ = fo.pcg_jacobi(A, b, rtol=1e-13)
    assert abs(it - it_o) <= 1 and np.abs(x - xo).max() < 1e-10 * np.abs(xo).max()


# ---- nonlinear Poisson + symmetric Nitsche (examples/nonlinear_poisson_opt) ----
def test_c_port_nonlinear_cycle_matches_the_numpy_oracle():
    """Round 5: the C/OpenMP port's nonlinear Poisson + Nitsche pieces (bench.py's config-5 CPU baseline at full size) against
    the NumPy oracle: residual and Jacobian to round-off on jittered 2-D / 3-D meshes, the whole SNES + adjoint cycle
    (BPX-CG where the oracle factorises) to solver accuracy."""
    from oracle import c_port as cp
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    for d, n in ((2, 16), (3, 6)):
        m = fo.unit_square_mesh(n, 0.2) if d == 2 else fo.unit_cube_mesh(n, 0.2)
        u, f = rng.standard_normal(m.n_vert), 1 + rng.random(m.n_cell)
        uex, bm = fo.u_exact_nl(m.x), fo.boundary_facets(m)
        for sgn, beta in ((1.0, fo.BETA_NITSCHE), (-1.0, 0.0)):
            R = fo.nl_residual(m, u, f, uex, bm, beta, sgn)
            Rc = cp.nl_residual(d, m.x, m.conn, u, f, uex, bm, beta, sgn)
            assert np.abs(R - Rc).max() <= 1e-13 * np.abs(R).max()
            J = fo.nl_jacobian(m, u, bm, beta, sgn).tocsr()
            rp, col = cp.pattern(d, m.n_vert, m.conn)
            Jc = sp.csr_matrix((cp.nl_jacobian(d, m.x, m.conn, u, bm, beta, rp, col, sgn), col, rp), shape=J.shape)
            assert abs(J - Jc).max() <= 1e-13 * abs(J).max()
    m = fo.unit_square_mesh(40)
    f = np.full(m.n_cell, 0.1)
    ref = fo.nl_reference_cycle(m, f, fo.u_exact_nl(m.x), fo.boundary_facets(m), fo.ALPHA_NL)
    out = cp.nl_cycle(2, m.x, m.conn, f, fo.u_exact_nl(m.x), fo.boundary_facets(m), fo.ALPHA_NL)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(out["u"], ref["u"]) <= 1e-9 and rel(out["grad"], ref["grad"]) <= 1e-9
    assert abs(out["J"] - ref["J"][0]) <= 1e-10 * abs(ref["J"][0]) and out["newton_its"] == ref["newton_its"]


@pytest.mark.parametrize("d,n", [(2, 8), (3, 4)])
def test_nl_oracle_consistency(d, n):
    m = fo.unit_square_mesh(n, 0.2) if d == 2 else fo.unit_cube_mesh(n, 0.2)
    bm = fo.boundary_facets(m)
    assert sum(bin(int(b)).count("1") for b in bm) == (4 * n if d == 2 else 12 * n * n)
    rng = np.random.default_rng(0)
    u, f = 0.5 * rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell)
    uex = fo.u_exact_nl(m.x)
    J = fo.nl_jacobian(m, u, bm)
    assert abs(J - J.T).max() < 1e-14                     # sym=True Nitsche: symmetric Jacobian
    du = rng.standard_normal(m.n_vert)
    fd = (fo.nl_residual(m, u + 1e-6 * du, f, uex, bm) - fo.nl_residual(m, u - 1e-6 * du, f, uex, bm)) / 2e-6
    assert np.abs(fd - J @ du).max() < 1e-8 * np.abs(J @ du).max()
    # cubic term: closed-form P1 monomial integrals vs brute-force tensor on one cell
    T3 = fo._p1_cubic_tables(d)
    assert np.isclose(T3.sum(), 1.0)                      # sum_abce int phi_a phi_b phi_c phi_e = |T|


def test_nl_unsymmetric_nitsche_oracle():
    """sym=False (run_nonlinear_poisson_opt.py:98-117): sgn = -1, no penalty -> non-symmetric Jacobian."""
    m = fo.unit_square_mesh(10, 0.2)
    bm = fo.boundary_facets(m)
    rng = np.random.default_rng(0)
    u, f, uex = 0.5 * rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell), fo.u_exact_nl(m.x)
    J = fo.nl_jacobian(m, u, bm, 0.0, -1.0)
    assert abs(J - J.T).max() > 0.1
    du = rng.standard_normal(m.n_vert)
    fd = (fo.nl_residual(m, u + 1e-6 * du, f, uex, bm, 0.0, -1.0) - fo.nl_residual(m, u - 1e-6 * du, f, uex, bm, 0.0, -1.0)) / 2e-6
    assert np.abs(fd - J @ du).max() < 1e-8 * np.abs(J @ du).max()
    out = fo.nl_reference_cycle(m, 0.1 * np.ones(m.n_cell), uex, bm, beta=0.0, sgn=-1.0)
    assert out["newton_its"] <= 6


def test_nl_manufactured_solution_converges():
    """u = sin 2pi x sin pi y, f = 5 pi^2 u + u^3 (run_nonlinear_poisson_opt.py:145-168): O(h^2) in L2."""
    errs = []
    for n in (8, 16, 32):
        m = fo.unit_square_mesh(n)
        bm = fo.boundary_facets(m)
        xc = fo.centroids(m)
        uc = np.sin(2 * np.pi * xc[:, 0]) * np.sin(np.pi * xc[:, 1])
        uex = fo.u_exact_nl(m.x)
        u, info = fo.nl_newton_solve(m, 5 * np.pi ** 2 * uc + uc ** 3, np.ones(m.n_vert), uex, bm)
        assert info.newton_its <= 6 and info.residual_norms[-1] < 1e-12
        errs.append(np.sqrt(2 * fo.functional(m, u, np.zeros(m.n_cell), uex, 0.0)))
    assert np.log2(errs[0] / errs[1]) > 1.7 and np.log2(errs[1] / errs[2]) > 1.85


def test_nl_golden_vector():
    g = np.load(os.path.join(GOLD, "nl_poisson_d2_n8.npz"))
    m = fo.OMesh(2, g["x"], g["conn"])
    assert np.array_equal(fo.boundary_facets(m), g["bmask"])
    ref = fo.nl_reference_cycle(m, g["f"], g["u_ex"], g["bmask"])
    for k in ("u", "J", "grad", "lam"):
        assert np.abs(ref[k] - g[k]).max() <= 1e-11 * np.abs(g[k]).max()
    assert int(ref["newton_its"]) == int(g["newton_its"])


# ---- Euler-Bernoulli beam (examples/beam_thickness_opt) ----
def test_beam_oracle_closed_form_and_reference_golden_vector():
    """Pins the beam oracle on (a) the closed-form tip deflection P L^3 / (3 E I) and (b) the golden
    vector held by the reference itself: the 50 optimal thicknesses of
    run_thickness_opt_cantilever_beam.py:252-261, reproduced by re-running SLSQP with the oracle's
    adjoint gradients."""
    import scipy.optimize as so
    nel, L, E, b, h = 50, 1.0, 1.0, 0.1, 0.1
    t0 = np.full(nel, h)
    c = fo.beam_cycle(nel, L, t0)
    assert abs(c["u"][2 * nel] + L ** 3 / (3 * E * b * h ** 3 / 12)) < 1e-8 * abs(c["u"][2 * nel])
    d = np.random.default_rng(0).standard_normal(nel) * 0.01
    fd = (fo.beam_cycle(nel, L, t0 + 1e-5 * d)["compliance"] - fo.beam_cycle(nel, L, t0 - 1e-5 * d)["compliance"]) / 2e-5
    assert abs(fd - c["grad_compliance"] @ d) < 1e-4 * abs(fd)
    res = so.minimize(lambda t: fo.beam_cycle(nel, L, t)["compliance"], t0, jac=lambda t: fo.beam_cycle(nel, L, t)["grad_compliance"],
                      bounds=[(1e-2, 10.)] * nel, method="SLSQP", options={"maxiter": 1000, "ftol": 1e-12},
                      constraints=[{"type": "eq", "fun": lambda t: fo.beam_cycle(nel, L, t)["volume"] - b * h * L,
                                    "jac": lambda t: fo.beam_cycle(nel, L, t)["grad_volume"]}])
    assert res.success and np.abs(res.x - fo.BEAM_THICK_REF).max() < 1e-6
