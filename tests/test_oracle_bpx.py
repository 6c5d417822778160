"""The NumPy restatement of the BPX preconditioner (oracle/bpx_oracle.py): lattice choice known
answers, symmetry / positive definiteness, and PCG with it against the direct solve."""
import numpy as np
import scipy.sparse.linalg as spla

from oracle import bpx_oracle as bo
from oracle import femo_oracle as fo


def _system(m, seed=0):
    bd = fo.boundary_vertices_box(m.x)
    A = fo.eliminate_bc(fo.stiffness(m), bd).tocsr()
    rng = np.random.default_rng(seed)
    b = fo.load_vector(m, 1.0 + rng.random(m.n_cell))
    b[bd] = 0.3 * rng.standard_normal(len(bd))
    pinned = np.zeros(m.n_vert, bool)
    pinned[bd] = True
    return A, b, pinned


def test_lattice_choice_known_answers():
    # the 10 M-DOF cube of the benchmark: 3 * 2^5 = 96 bins, 6 levels, 97^3 nodes on the finest one
    bins, H = bo.choose_lattice(np.zeros(3), np.ones(3), 216 ** 3)
    assert len(bins) == 6 and list(bins[-1]) == [96, 96, 96] and list(bins[0]) == [3, 3, 3]
    assert np.prod(bins[-1] + 1) == 912673 and abs(H[-1] - 1.0 / 96) < 1e-15
    # BASELINE config 5: the 2236^2 square
    bins, H = bo.choose_lattice(np.zeros(2), np.ones(2), 2237 ** 2)
    assert list(bins[-1]) == [1024, 1024] and len(bins) == 10
    # anisotropic box: bins follow the extents (at least one per axis), every level doubles all axes
    bins, H = bo.choose_lattice(np.zeros(3), np.array([2.0, 1.0, 0.5]), 129 * 65 * 33)
    assert len(bins) == 6 and all(list(b) == [2 << l, 1 << l, 1 << l] for l, b in enumerate(bins))
    assert H[-1] == 2.0 / 64


def test_operator_is_spd_and_nested_transfers_compose():
    m = fo.unit_cube_mesh(6, 0.2)
    A, b, pinned = _system(m)
    M = bo.BPX(m.x, A.diagonal(), pinned)
    Minv = M.matrix()
    assert np.abs(Minv - Minv.T).max() < 1e-13 * np.abs(Minv).max()
    assert np.linalg.eigvalsh(0.5 * (Minv + Minv.T)).min() > 0.0
    # pinned vertices only see the Jacobi term
    r = np.random.default_rng(1).standard_normal(m.n_vert)
    assert np.array_equal(M.apply(r)[pinned], (M.dinv * r)[pinned])
    # P_l = P_L I ... I reproduces multilinear interpolation from the coarse lattice directly
    # (exactly, for vertices whose fractions are representable: use the unquantised operator)
    P_fine = bo.interpolation(m.x, M.lo, M.hi, M.bins[-1], quantise=False)
    P_coarse = bo.interpolation(m.x, M.lo, M.hi, M.bins[-2], quantise=False)
    assert abs(P_fine @ M.I[-1] - P_coarse).max() < 1e-13


def test_pcg_matches_direct_solve_with_few_iterations():
    its = {}
    for d, n, jit in [(2, 64, 0.2), (3, 12, 0.0), (3, 16, 0.25), (3, 24, 0.2)]:
        m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
        A, b, pinned = _system(m, seed=n)
        M = bo.BPX(m.x, A.diagonal(), pinned)
        x, it = bo.pcg(A, b, M, rtol=1e-14)
        x_ref = spla.spsolve(A.tocsc(), b)
        assert np.abs(x - x_ref).max() < 1e-10 * np.abs(x_ref).max()
        _, it_jacobi, _ = fo.pcg_jacobi(A, b, rtol=1e-14)
        assert it < it_jacobi and it <= 60
        its[(d, n)] = it
    assert its[(3, 24)] <= its[(3, 12)] + 12, its


def test_weak_boundary_conditions_pin_the_facet_vertices():
    m = fo.unit_square_mesh(32, 0.1)
    bm = fo.boundary_facets(m)
    u = 0.3 * np.sin(3 * m.x[:, 0]) + 0.2
    J = fo.nl_jacobian(m, u, bm).tocsr()
    pinned = np.zeros(m.n_vert, bool)
    pinned[fo.boundary_vertices_box(m.x)] = True
    b = np.random.default_rng(0).standard_normal(m.n_vert)
    x, it = bo.pcg(J, b, bo.BPX(m.x, J.diagonal(), pinned), rtol=1e-13)
    x_ref = spla.spsolve(J.tocsc(), b)
    assert np.abs(x - x_ref).max() < 1e-9 * np.abs(x_ref).max() and it <= 60


def test_c_port_bpx_matches_the_numpy_operator_and_solves():
    """oracle/femo_oracle_c.c (the CPU-baseline port) applies the same operator and its BPX-PCG
    reproduces the cycle of the Jacobi-CG port."""
    from oracle import c_port
    m = fo.unit_cube_mesh(14, 0.2)
    A, b, pinned = _system(m)
    M = bo.BPX(m.x, A.diagonal(), pinned)
    B = c_port.Bpx(m.x, pinned)
    assert B.levels == M.levels
    r = np.random.default_rng(5).standard_normal(m.n_vert)
    z = B.apply(M.dinv, r)
    ref = M.apply(r)
    assert np.abs(z - ref).max() < 1e-12 * np.abs(ref).max()
    rowptr, col = c_port.pattern(3, m.n_vert, m.conn)
    x, it, res = c_port.pcg_bpx(B, rowptr, col, A.data, b, rtol=1e-14)
    x_np, it_np = bo.pcg(A, b, M, rtol=1e-14)
    assert abs(it - it_np) <= 1 and np.abs(x - x_np).max() < 1e-11 * np.abs(x_np).max()
    # whole cycle: same state / gradient with either preconditioner, far fewer iterations with BPX
    bd = fo.boundary_vertices_box(m.x)
    f = 0.086 * (1.0 + 0.3 * np.random.default_rng(11).uniform(-1, 1, m.n_cell))
    cj = c_port.poisson_cycle(3, m.x, m.conn, f, fo.u_target(m.x), bd, 1e-6, pc="jacobi")
    cb = c_port.poisson_cycle(3, m.x, m.conn, f, fo.u_target(m.x), bd, 1e-6, pc="bpx")
    assert np.abs(cb["u"] - cj["u"]).max() < 1e-10 * np.abs(cj["u"]).max()
    assert np.abs(cb["grad"] - cj["grad"]).max() < 1e-10 * np.abs(cj["grad"]).max()
    assert cb["it_adj"] < cj["it_adj"]
