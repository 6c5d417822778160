"""Hermite-type lattice spaces in the shell's preconditioner (round 4): csrc/shell.hip (k_pc_restrict_h, k_lat_level_h,
k_lat_down_composite_h, the Hermite branch of the prolongations, k_pc_galerkin_blocks_h, k_pc_coarse_galerkin_h) through
femo_amd/fea/shell.py against oracle/shell_oracle.py::LatticePreconditioner -- the dense coarse operator entry by entry,
M^-1 r, the iteration counts, and the solution against the direct solve."""
import numpy as np
import pytest

from oracle import shell_oracle as so
from tests.test_gpu_shell import E_ROOF, FZ, H_ROOF, NU_ROOF, rel, roof_fixed

pytestmark = pytest.mark.gpu


def _problem(ctx, n, hermite=True, coarse=None, finest=None):
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(n, n)
    V0 = so.ShellSpace(pts, conn)
    fixed = roof_fixed(V0)
    prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=fixed, ctx=ctx, pc="lattice")
    prob.dev.enable_lattice_pc(finest=finest, hermite=hermite, coarse_unknowns=coarse)
    prob.set_thickness(H_ROOF)
    prob.set_load([0.0, 0.0, FZ])
    return prob, V0, fixed


@pytest.mark.parametrize("n,coarse", [(16, None), (16, 300), (24, None)])
def test_hermite_operator_matches_the_oracle(ctx, n, coarse):
    """The dense Galerkin operator of the coarse-solve level (composed Hermite prolongation) and M^-1 r as a whole: the
    kernels apply the operator the oracle writes down.  coarse = 300 puts two levels of node blocks and the composite
    restriction between the finest lattice and the coarse solve."""
    from femo_amd.engine import Vec
    prob, V0, fixed = _problem(ctx, n, coarse=coarse)
    assert prob.dev.hermite
    K = so.assemble(V0, so.element_stiffness(V0, np.full(V0.n_vert, H_ROOF), E_ROOF, NU_ROOF)).tocsr()
    M = so.LatticePreconditioner(V0, K, fixed, hermite=True, coarse_max=coarse or 3200)
    assert M.c == prob.dev.coarse_level and M.levels == prob.dev.pc_levels
    mask = np.zeros(V0.n_dof, dtype=np.uint8)
    mask[fixed] = 1
    vals = prob._stiffness()
    A = prob.dev.coarse_matrix(vals, mask)
    A_ref = (M.P[M.c].T @ M.Kf @ M.P[M.c]).toarray()
    # (the oracle's P has zero rows on the fixed dofs, so the identity rows of Kf do not enter)
    assert A.shape == A_ref.shape
    assert np.abs(A - A_ref).max() <= 2e-6 * np.abs(A_ref).max()          # transfer weights are single precision on the device
    assert np.abs(A - A.T).max() <= 1e-12 * np.abs(A).max()
    rng = np.random.default_rng(3)
    r = rng.standard_normal(V0.n_dof)
    z = prob.dev.pc_apply(vals, Vec(ctx, V0.n_dof).set(r), Vec(ctx, V0.n_dof), mask).get()
    st = prob.dev.pc_state()                # after the first apply: the coarse operator is factorised, nothing fell back
    assert st["hermite_enabled"] and st["coarse_solve_ready"] and not st["fell_back_to_trilinear"], st
    z_ref = M.apply(r)
    assert np.abs(z - z_ref).max() <= 2e-5 * np.abs(z_ref).max()
    # symmetric positive definite: <r1, M^-1 r2> = <M^-1 r1, r2>, <r, M^-1 r> > 0
    r2 = rng.standard_normal(V0.n_dof)
    z2 = prob.dev.pc_apply(vals, Vec(ctx, V0.n_dof).set(r2), Vec(ctx, V0.n_dof), mask).get()
    free = mask == 0
    assert abs(r[free] @ z2[free] - z[free] @ r2[free]) <= 1e-6 * abs(r[free] @ z2[free])
    assert r[free] @ z[free] > 0.0


def test_hermite_iteration_counts(ctx):
    """The counts of the oracle's PCG with the same operator are reproduced (16 x 16: 153, 32 x 32: 85 -- pinned in
    tests/test_oracle_shell.py), they are well below the trilinear hierarchy's (258 / 149), and the solution is the direct one."""
    res = {}
    for n in (16, 32):
        for herm in (False, True):
            prob, V0, fixed = _problem(ctx, n, hermite=herm)
            w = prob.solve(rtol=1e-10)
            res[(n, herm)] = (prob.last_info.iterations, prob.last_info.converged, w)
        K = so.assemble(V0, so.element_stiffness(V0, np.full(V0.n_vert, H_ROOF), E_ROOF, NU_ROOF)).tocsr()
        F = so.load_vector(V0, np.tile([0.0, 0.0, FZ], (V0.n_vert, 1)))
        w_ref = so.solve(K, F, fixed)
        for herm in (False, True):
            assert res[(n, herm)][1] == 1
            assert rel(res[(n, herm)][2], w_ref) <= 1e-7
    assert abs(res[(16, True)][0] - 153) <= 6 and abs(res[(32, True)][0] - 85) <= 4
    assert res[(16, True)][0] < 0.7 * res[(16, False)][0] and res[(32, True)][0] < 0.65 * res[(32, False)][0]


@pytest.mark.parametrize("n,coarse", [(16, None), (24, 300)])
def test_node_blocks_by_items_equal_the_row_wise_kernel(ctx, n, coarse, monkeypatch):
    """The node blocks of the levels above the coarse solve formed item by item (a wave per lattice cell,
    k_pc_galerkin_blocks_w) against the row-wise kernel (k_pc_galerkin_blocks_h, FEMO_SHELL_BLOCKS_BY_ROWS): the same sums in
    another order, so M^-1 r agrees to rounding -- with the Dirichlet mask of the roof and without a mask."""
    from femo_amd.engine import Vec
    rng = np.random.default_rng(5)
    for with_mask in (True, False):
        zs = []
        for by_rows in (True, False):
            if by_rows:
                monkeypatch.setenv("FEMO_SHELL_BLOCKS_BY_ROWS", "1")
            else:
                monkeypatch.delenv("FEMO_SHELL_BLOCKS_BY_ROWS", raising=False)
            prob, V0, fixed = _problem(ctx, n, coarse=coarse)
            mask = np.zeros(V0.n_dof, dtype=np.uint8)
            if with_mask:
                mask[fixed] = 1
            r = np.random.default_rng(7).standard_normal(V0.n_dof)
            zs.append(prob.dev.pc_apply(prob._stiffness(), Vec(ctx, V0.n_dof).set(r), Vec(ctx, V0.n_dof), mask).get())
        assert np.abs(zs[0] - zs[1]).max() <= 1e-5 * np.abs(zs[0]).max()      # measured 2e-9 / 5e-7: the sums' order through the blocks' conditioning


def test_node_blocks_fall_back_when_elements_span_lattice_cells(ctx, monkeypatch):
    """A finest lattice finer than the mesh (8 x 8 roof, 64 cells per axis): column points lie more than one cell away, the
    cell-per-wave kernel flags it on the device and the set-up takes the row-wise kernel -- same M^-1 r as when the row-wise
    kernel is asked for, and the solve still reaches the direct solution."""
    from femo_amd.engine import Vec
    zs = []
    for by_rows in (True, False):
        if by_rows:
            monkeypatch.setenv("FEMO_SHELL_BLOCKS_BY_ROWS", "1")
        else:
            monkeypatch.delenv("FEMO_SHELL_BLOCKS_BY_ROWS", raising=False)
        prob, V0, fixed = _problem(ctx, 8, finest=64)
        mask = np.zeros(V0.n_dof, dtype=np.uint8)
        mask[fixed] = 1
        r = np.random.default_rng(11).standard_normal(V0.n_dof)
        zs.append(prob.dev.pc_apply(prob._stiffness(), Vec(ctx, V0.n_dof).set(r), Vec(ctx, V0.n_dof), mask).get())
    assert np.abs(zs[0] - zs[1]).max() <= 1e-9 * np.abs(zs[0]).max()
    w = prob.solve(rtol=1e-10)
    K = so.assemble(V0, so.element_stiffness(V0, np.full(V0.n_vert, H_ROOF), E_ROOF, NU_ROOF)).tocsr()
    F = so.load_vector(V0, np.tile([0.0, 0.0, FZ], (V0.n_vert, 1)))
    assert prob.last_info.converged in (1, 2) and rel(w, so.solve(K, F, fixed)) <= 1e-6


def test_set_up_kernels_on_a_high_valence_mesh(ctx, monkeypatch):
    """A fan of 16 triangles around one vertex (valence 16: 66 node blocks in that vertex's row, three staging rounds of the
    cell-per-wave kernel, more than 64 blocks of one point in the matrix-core kernel's accumulation): the default set-up kernels
    against the row-wise node-block kernel and the LDS-atomic dense Galerkin kernel, and the solve against the oracle."""
    from femo_amd.engine import Vec
    from femo_amd.fea.shell import ShellProblem
    ns = 16
    ang = 2.0 * np.pi * np.arange(ns) / ns
    ring = lambda rad, z: np.stack([rad * np.cos(ang), rad * np.sin(ang), np.full(ns, z)], axis=1)
    pts = np.concatenate([[[0.0, 0.0, 0.3]], ring(1.0, 0.2), ring(2.0, 0.0)])                # a shallow cap
    tri = []
    for k in range(ns):
        k1 = (k + 1) % ns
        tri.append([0, 1 + k, 1 + k1])
        tri.append([1 + k, 1 + ns + k, 1 + ns + k1])
        tri.append([1 + k, 1 + ns + k1, 1 + k1])
    conn = np.asarray(tri, dtype=np.int32)
    V0 = so.ShellSpace(pts, conn)
    outer = np.arange(1 + ns, 1 + 2 * ns)
    on_rim = np.flatnonzero(np.isclose(np.hypot(V0.unode_x[:, 0], V0.unode_x[:, 1]), 2.0, atol=0.05) | (np.hypot(V0.unode_x[:, 0], V0.unode_x[:, 1]) > 1.9))
    fixed = np.unique(np.concatenate([V0.u_dof(on_rim, c) for c in range(3)] + [V0.theta_dof(outer, c) for c in range(3)]))
    h = np.full(V0.n_vert, 0.05)
    mask = np.zeros(V0.n_dof, dtype=np.uint8)
    mask[fixed] = 1
    r = np.random.default_rng(13).standard_normal(V0.n_dof)
    zs = {}
    for rows, atomic in ((True, True), (False, False)):
        for name, on in (("FEMO_SHELL_BLOCKS_BY_ROWS", rows), ("FEMO_SHELL_CG_ATOMIC", atomic)):
            if on:
                monkeypatch.setenv(name, "1")
            else:
                monkeypatch.delenv(name, raising=False)
        prob = ShellProblem(pts, conn, 2.0e8, 0.3, fixed_dofs=fixed, ctx=ctx, pc="lattice")
        prob.dev.enable_lattice_pc(coarse_unknowns=400)
        assert prob.dev.hermite and prob.dev.coarse_level is not None
        prob.set_thickness(h)
        prob.set_load([0.0, 0.0, -1.0e3])
        zs[(rows, atomic)] = prob.dev.pc_apply(prob._stiffness(), Vec(ctx, V0.n_dof).set(r), Vec(ctx, V0.n_dof), mask).get()
    a, b = zs[(True, True)], zs[(False, False)]
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()
    w = prob.solve(rtol=1e-11)
    K = so.assemble(V0, so.element_stiffness(V0, h, 2.0e8, 0.3)).tocsr()
    F = so.load_vector(V0, np.tile([0.0, 0.0, -1.0e3], (V0.n_vert, 1)))
    assert prob.last_info.converged in (1, 2) and rel(w, so.solve(K, F, fixed)) <= 1e-6


def test_dirichlet_mask_kept_on_the_device_follows_the_callers_array(ctx):
    """Round 5: the shell keeps the Dirichlet mask of the last solve on the device behind a hash of the caller's array
    (`shell_mask`, shell.hip) instead of uploading it per solve.  Another mask must replace it -- state, preconditioner
    set-up and imposed dofs all follow --, and coming back to the first mask gives the first result again."""
    from femo_amd.engine import Vec
    prob, V0, fixed = _problem(ctx, 16)
    vals = prob._stiffness()
    prob.dev.load(prob.f, prob.F)
    n = V0.n_dof
    mask_a = np.zeros(n, dtype=np.uint8)
    mask_a[fixed] = 1
    extra = np.setdiff1d(np.arange(0, n, 97), fixed)[:25]          # pin two dozen more dofs: another problem
    mask_b = mask_a.copy()
    mask_b[extra] = 1
    out = []
    for mask in (mask_a, mask_b, mask_a, mask_b.copy()):          # (the last one: equal content in another array)
        x = Vec(ctx, n)
        info = prob.dev.solve(vals, prob.F, x, fixed=mask, rtol=1e-11, pc="lattice")
        assert info.converged in (1, 2)
        out.append(np.array(x.get()))
    xa, xb, xa2, xb2 = out
    assert np.all(xa[fixed] == 0.0) and np.all(xb[fixed] == 0.0) and np.all(xb[extra] == 0.0)
    assert np.abs(xa[extra]).max() > 0.0                           # ... which were free under the first mask
    assert rel(xa2, xa) < 1e-9 and rel(xb2, xb) < 1e-9             # (two PCG runs: equal to the solver tolerance)
    assert rel(xb, xa) > 1e-3                                      # and the two problems do differ


def test_results_are_the_callers_own_arrays(ctx):
    """`ShellProblem` hands out the pinned blocks its results landed in as writable arrays (round 5: no pageable copy):
    the caller may write into them, and a later call neither sees that nor overwrites what was handed out."""
    prob, V0, fixed = _problem(ctx, 16)
    w1 = prob.solve(rtol=1e-11)
    assert w1.flags.writeable and w1.dtype == np.float64 and w1.size == V0.n_dof
    keep = w1.copy()
    J1, dJdw = prob.compliance(grad=True)
    dJdw[fixed] = 0.0                                              # (what the operators do with it)
    w1 *= 2.0                                                      # the caller's array now
    w2 = prob.solve(rtol=1e-11)
    assert w2 is not w1 and w2.ctypes.data != w1.ctypes.data
    assert rel(w2, keep) < 1e-9 and np.array_equal(w1, 2.0 * keep)
    J2, _ = prob.compliance(grad=True)
    assert abs(J2 - J1) <= 1e-9 * abs(J1)
