"""Reissner-Mindlin shell kernels (csrc/shell.hip through femo_amd/fea/shell.py) against oracle/shell_oracle.py:
element couplings, load, residual, forward / adjoint solves on the Scordelis-Lo roof, thickness sensitivities."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import shell_oracle as so

pytestmark = pytest.mark.gpu

E_ROOF, NU_ROOF, H_ROOF, FZ, L = 4.32e8, 0.0, 0.25, -90.0, 25.0


def roof_fixed(V):
    ux, vx = V.unode_x, V.x
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    return np.unique(np.concatenate([
        V.u_dof(on(ux[:, 0], L), 1), V.u_dof(on(ux[:, 0], L), 2),
        V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0), V.theta_dof(on(vx[:, 1], 0.0), 2),
        V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


@pytest.fixture(scope="module")
def roof(ctx):
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(8, 8)
    rng = np.random.default_rng(0)
    pts = pts.copy()
    V0 = so.ShellSpace(pts, conn)
    prob = ShellProblem(pts, conn, E_ROOF, 0.3, fixed_dofs=roof_fixed(V0), ctx=ctx)
    h = H_ROOF * (1.0 + 0.3 * rng.random(V0.n_vert))
    f = np.tile([0.0, 0.0, FZ], (V0.n_vert, 1)) * (1.0 + 0.2 * rng.random((V0.n_vert, 1)))
    prob.set_thickness(h)
    prob.set_load(f)
    return prob, V0, h, f


def test_space_matches_the_oracle(roof):
    prob, V0, _, _ = roof
    S = prob.space
    assert S.n_dof == V0.n_dof and np.array_equal(S.cell_dofs, V0.cell_dofs) and np.array_equal(S.cell_edges, V0.cell_edges)
    rowptr, cols, epos = S.pattern()
    assert rowptr[-1] == cols.size and epos.shape == (S.n_cell, 729)
    # positions point at the right (row, col)
    rows = np.repeat(np.arange(S.n_dof), np.diff(rowptr))
    c = 5
    assert np.array_equal(rows[epos[c]].reshape(27, 27), np.repeat(S.cell_dofs[c], 27).reshape(27, 27))
    assert np.array_equal(cols[epos[c]].reshape(27, 27), np.tile(S.cell_dofs[c], 27).reshape(27, 27))


def test_stiffness_load_and_residual(roof):
    prob, V0, h, f = roof
    S = prob.space
    rowptr, cols, _ = S.pattern()
    vals = np.array(prob._stiffness().get())
    K = sp.csr_matrix((vals, cols, rowptr), shape=(S.n_dof, S.n_dof))
    Kref = so.assemble(V0, so.element_stiffness(V0, h, E_ROOF, 0.3))
    assert abs(K - Kref).max() <= 1e-12 * abs(Kref).max()
    assert abs(K - K.T).max() <= 1e-12 * abs(Kref).max()
    Fref = so.load_vector(V0, f)
    prob.dev.load(prob.f, prob.F)
    assert rel(prob.F.get(), Fref) <= 1e-13
    w = np.random.default_rng(1).standard_normal(S.n_dof)
    assert rel(prob.residual(w), Kref @ w - Fref) <= 1e-12


def test_forward_and_adjoint_solves(roof):
    prob, V0, h, f = roof
    Kref = so.assemble(V0, so.element_stiffness(V0, h, E_ROOF, 0.3))
    Fref = so.load_vector(V0, f)
    fixed = np.nonzero(prob.fixed)[0]
    wref = so.solve(Kref, Fref, fixed)
    w = prob.solve()
    assert prob.last_info.converged in (1, 2) and prob.last_info.iterations > 100
    # the attainable bound: eps * cond(K_ff) limits what ANY fp64 solver delivers in the solution -- the oracle's SuperLU
    # included -- so two correct solutions agree to a small multiple of it (cond = 2.3e6 on this 8 x 8 roof: bound 5e-10)
    free = np.setdiff1d(np.arange(V0.n_dof), fixed)
    ev = np.linalg.eigvalsh(Kref[free][:, free].toarray())
    bound = np.finfo(float).eps * ev[-1] / ev[0]
    assert 1e-10 < bound < 1e-8
    print(f"shell solve: error {rel(w, wref):.2e}, eps*cond {bound:.2e}")
    assert rel(w, wref) <= bound                                     # measured: 1.5e-11
    assert np.all(w[fixed] == 0.0)
    c = np.random.default_rng(2).standard_normal(V0.n_dof)
    c[fixed] = 0.0
    lam = prob.solve_adjoint(c)
    print(f"shell adjoint solve: error {rel(lam, so.solve(Kref, c, fixed)):.2e}")
    assert rel(lam, so.solve(Kref, c, fixed)) <= bound
    # adjoint identity <K^-1 F, c> = <F, K^-T c> on the free dofs
    Ff = Fref.copy(); Ff[fixed] = 0.0
    assert abs(w @ c - Ff @ lam) <= 1e-9 * abs(w @ c)


def test_outputs_and_their_partials(roof):
    prob, V0, h, f = roof
    w = np.random.default_rng(3).standard_normal(V0.n_dof) * 1e-3
    J, g = prob.compliance(w, grad=True)
    assert J == pytest.approx(so.compliance(V0, w), rel=1e-12)
    d = np.random.default_rng(4).standard_normal(V0.n_dof) * 1e-3
    fd = (so.compliance(V0, w + 1e-4 * d) - so.compliance(V0, w - 1e-4 * d)) / 2e-4
    assert g @ d == pytest.approx(fd, rel=1e-8)
    M, gm = prob.mass(2.5, grad=True)
    _, _, _, area, _ = V0.frames()
    assert M == pytest.approx(2.5 * float((area[:, None] / 3.0 * h[V0.conn]).sum()), rel=1e-13)
    assert gm.sum() == pytest.approx(2.5 * area.sum(), rel=1e-13)
    en = prob.elastic_energy(w)
    assert en == pytest.approx(sum(so.energy_parts(V0, w, h, E_ROOF, 0.3).values()), rel=1e-11)


def test_pnorm_stress_and_its_partials(roof):
    """`pnorm_stress` (shell_pde.py:297-313) against the oracle: value, dJ/dw, dJ/dh on the three surfaces; and its total
    derivative w.r.t. the thickness through the adjoint against differences of the oracle's direct solves."""
    prob, V0, h, f = roof
    rng = np.random.default_rng(21)
    w = 1e-3 * rng.standard_normal(V0.n_dof)
    for surface, m, rho in ((1.0, 2e-6, 6.0), (-1.0, 2e-6, 6.0), (0.0, 1e-5, 3.0), (1.0, 1e-6, 100.0)):
        J, gw, gh = prob.pnorm_stress(w, m=m, rho=rho, surface=surface, grad=True)
        Jr, gwr, ghr = so.pnorm_stress(V0, w, h, E_ROOF, 0.3, m=m, rho=rho, surface=surface, grad=True)
        assert J == pytest.approx(Jr, rel=1e-11)
        assert np.abs(gw - gwr).max() <= 1e-10 * np.abs(gwr).max()
        assert np.abs(gh - ghr).max() <= 1e-10 * np.abs(ghr).max()
        assert prob.pnorm_stress(w, m=m, rho=rho, surface=surface) == pytest.approx(Jr, rel=1e-11)
    # total derivative dJ/dh = partial - lam^T dK/dh w with K lam = dJ/dw
    m, rho = 2e-6, 4.0
    fixed = np.nonzero(prob.fixed)[0]
    wsol = prob.solve(rtol=1e-12)
    J, gw, gh = prob.pnorm_stress(m=m, rho=rho, grad=True)
    gw[fixed] = 0.0
    lam = prob.solve_adjoint(gw, rtol=1e-12)
    total = gh - prob.dRdh_T(lam, wsol)

    def Jref(hh):
        K = so.assemble(V0, so.element_stiffness(V0, hh, E_ROOF, 0.3))
        return so.pnorm_stress(V0, so.solve(K, so.load_vector(V0, f), fixed), hh, E_ROOF, 0.3, m=m, rho=rho)

    dh = 0.01 * rng.standard_normal(V0.n_vert)
    assert total @ dh == pytest.approx((Jref(h + 1e-2 * dh) - Jref(h - 1e-2 * dh)) / 2e-2, rel=1e-4)
    prob.set_thickness(h)


def test_von_mises_field(roof):
    """`von_Mises_stress` + `projected_von_Mises_stress` (shell_pde.py:315-332): consistent and lumped L2 projection onto
    the vertices against the oracle, on the three surfaces."""
    prob, V0, h, f = roof
    w = 1e-3 * np.random.default_rng(31).standard_normal(V0.n_dof)
    for surface in (1.0, 0.0, -1.0):
        for lump in (False, True):
            got = prob.von_mises_field(w, surface=surface, lump_mass=lump)
            ref = so.project_von_mises(V0, w, h, E_ROOF, 0.3, surface, lump_mass=lump)
            assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


def test_thickness_derivative_of_the_bilinear_form(roof):
    prob, V0, h, f = roof
    rng = np.random.default_rng(5)
    w, lam = rng.standard_normal(V0.n_dof), rng.standard_normal(V0.n_dof)
    g = prob.dRdh_T(lam, w)
    dh = rng.standard_normal(V0.n_vert) * 0.01

    def form(hh):
        Ke = so.element_stiffness(V0, hh, E_ROOF, 0.3)
        return float(np.einsum("ca,cab,cb->", lam[V0.cell_dofs], Ke, w[V0.cell_dofs]))

    fd = (form(h + 1e-3 * dh) - form(h - 1e-3 * dh)) / 2e-3
    assert g @ dh == pytest.approx(fd, rel=1e-6)
    gf = prob.dRdf_T(lam)
    df = rng.standard_normal((V0.n_vert, 3))
    assert (gf * df).sum() == pytest.approx(-lam @ so.load_vector(V0, df), rel=1e-12)


def test_compliance_gradient_by_the_adjoint(ctx):
    """BASELINE config 3 in miniature: thickness sensitivity of an output through the adjoint of the shell solve,
    against central differences of the oracle's direct solves."""
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(4, 4)
    V0 = so.ShellSpace(pts, conn)
    fixed = roof_fixed(V0)
    prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=fixed, ctx=ctx)
    rng = np.random.default_rng(6)
    h = H_ROOF * (1.0 + 0.2 * rng.random(V0.n_vert))
    f = np.tile([0.0, 0.0, FZ], (V0.n_vert, 1))
    prob.set_thickness(h); prob.set_load(f)
    J, g, w = prob.compliance_gradient()

    def Jref(hh):
        K = so.assemble(V0, so.element_stiffness(V0, hh, E_ROOF, NU_ROOF))
        return so.compliance(V0, so.solve(K, so.load_vector(V0, f), fixed))

    assert J == pytest.approx(Jref(h), rel=1e-7)
    dh = rng.standard_normal(V0.n_vert) * 0.01
    fd = (Jref(h + 1e-2 * dh) - Jref(h - 1e-2 * dh)) / 2e-2
    assert g @ dh == pytest.approx(fd, rel=1e-4)


def test_scordelis_lo_on_the_gpu(ctx):
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(16, 16)
    V0 = so.ShellSpace(pts, conn)
    prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=roof_fixed(V0), ctx=ctx)
    prob.set_thickness(H_ROOF)
    prob.set_load([0.0, 0.0, FZ])
    w = prob.solve(rtol=1e-10)
    tip = int(np.argmin(np.abs(V0.x[:, 0]) + np.abs(V0.x[:, 1] - V0.x[:, 1].max())))
    ref, _, _ = so.scordelis_lo(16, 16)
    assert V0.vertex_displacement(w)[tip, 2] == pytest.approx(ref, rel=1e-6)
    assert V0.vertex_displacement(w)[tip, 2] == pytest.approx(-0.3024, rel=0.015)       # run_shape_opt_roof.py:224


def test_shell_through_the_operator_stack(ctx):
    """ShellPDE forms behind FEA / FEAModel / StateOperation / OutputOperation (shell_module.py:20-120): run and the
    reverse sweep of the in-repo driver, against the oracle's direct solves (J) and central differences (dJ/dh)."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA
    from femo_amd.fea.function import Function
    from femo_amd.fea.shell_forms import ShellMesh, ShellPDE, locate_shell_dofs
    utils_hip.set_context(ctx)
    pts, conn = so.scordelis_lo_mesh(4, 4)
    V0 = so.ShellSpace(pts, conn)
    mesh = ShellMesh(pts, conn)
    pde = ShellPDE(mesh)
    fea = FEA(mesh)
    fea.REPORT = False
    fea.linear_problem = True
    h_fn, f_fn, w_fn = Function(pde.VT), Function(pde.VF), Function(pde.W)
    res = pde.pdeRes(h_fn, w_fn, f_fn, E_ROOF, NU_ROOF)
    fea.add_input('thickness', h_fn, init_val=H_ROOF)
    fea.add_input('F_solid', f_fn, init_val=0.0)
    fea.add_state(name='disp_solid', function=w_fn, residual_form=res, arguments=['thickness', 'F_solid'])
    fea.add_output(name='compliance', type='scalar', form=pde.compliance(w_fn, h_fn), arguments=['disp_solid', 'thickness'])
    fea.add_output(name='mass', type='scalar', form=pde.mass(h_fn, 2.0), arguments=['thickness'])
    fea.add_output(name='elastic_energy', type='scalar', form=pde.elastic_energy(w_fn, h_fn, E_ROOF, NU_ROOF),
                   arguments=['disp_solid', 'thickness'])
    fea.add_output(name='pnorm_stress', type='scalar', form=pde.pnorm_stress(w_fn, h_fn, E_ROOF, NU_ROOF, m=2e-6, rho=4),
                   arguments=['disp_solid', 'thickness'])
    fea.add_field_output(name='von_Mises_stress', form=pde.von_Mises_stress(w_fn, h_fn, E_ROOF, NU_ROOF, surface='Top'),
                         arguments=['disp_solid', 'thickness'])                         # shell_dynamic_pde.py:82-83,129
    at = lambda k, v: (lambda x: np.isclose(x[k], v, atol=1e-6))
    ubc = Function(pde.W)
    ubc.vector.set(0.0)
    locs = [locate_shell_dofs(pde.W, 'u', 1, at(0, L)), locate_shell_dofs(pde.W, 'u', 2, at(0, L)),
            locate_shell_dofs(pde.W, 'u', 1, at(1, 0.0)), locate_shell_dofs(pde.W, 'theta', 0, at(1, 0.0)),
            locate_shell_dofs(pde.W, 'theta', 2, at(1, 0.0)), locate_shell_dofs(pde.W, 'u', 0, at(0, 0.0)),
            locate_shell_dofs(pde.W, 'theta', 1, at(0, 0.0)), locate_shell_dofs(pde.W, 'theta', 2, at(0, 0.0))]
    fea.add_strong_bc(ubc, locs, pde.W)
    fixed = np.unique(np.concatenate(locs))
    assert np.array_equal(fixed, roof_fixed(V0))
    model = FEAModel(fea=[fea])
    rng = np.random.default_rng(8)
    h = H_ROOF * (1.0 + 0.2 * rng.random(V0.n_vert))
    f = np.tile([0.0, 0.0, FZ], (V0.n_vert, 1)).ravel()
    model.create_input('thickness', shape=V0.n_vert, val=h)
    model.create_input('F_solid', shape=3 * V0.n_vert, val=f)
    sim = Simulator(model)
    sim.run()

    def solve_ref(hh, ff=f):
        K = so.assemble(V0, so.element_stiffness(V0, hh, E_ROOF, NU_ROOF))
        return so.solve(K, so.load_vector(V0, ff.reshape(-1, 3)), fixed)

    wref = solve_ref(h)
    assert rel(sim['disp_solid'], wref) <= 1e-9                      # eps * cond(K_ff) of this roof, see test_forward_and_adjoint_solves
    assert sim['compliance'][0] == pytest.approx(so.compliance(V0, wref), rel=1e-8)
    assert sim['elastic_energy'][0] == pytest.approx(sum(so.energy_parts(V0, wref, h, E_ROOF, NU_ROOF).values()), rel=1e-8)
    _, _, _, area, _ = V0.frames()
    assert sim['mass'][0] == pytest.approx(2.0 * float((area[:, None] / 3.0 * h[V0.conn]).sum()), rel=1e-12)
    # thickness sensitivity of the compliance through the adjoint of the shell solve
    g = np.asarray(sim.compute_totals('compliance', 'thickness'))
    dh = rng.standard_normal(V0.n_vert) * 0.01
    Jref = lambda hh: so.compliance(V0, solve_ref(hh))
    fd = (Jref(h + 1e-2 * dh) - Jref(h - 1e-2 * dh)) / 2e-2
    assert g @ dh == pytest.approx(fd, rel=1e-4)
    # ... and w.r.t. the nodal forces
    gf = np.asarray(sim.compute_totals('compliance', 'F_solid'))
    df = rng.standard_normal(3 * V0.n_vert)
    Jf = lambda ff: so.compliance(V0, solve_ref(h, ff))
    fdf = (Jf(f + 1e-3 * df) - Jf(f - 1e-3 * df)) / 2e-3
    assert gf @ df == pytest.approx(fdf, rel=1e-6)
    # the energy output has partials w.r.t. state and thickness: total derivative against differences
    ge = np.asarray(sim.compute_totals('elastic_energy', 'thickness'))
    Eref = lambda hh: sum(so.energy_parts(V0, solve_ref(hh), hh, E_ROOF, NU_ROOF).values())
    fde = (Eref(h + 1e-2 * dh) - Eref(h - 1e-2 * dh)) / 2e-2
    assert ge @ dh == pytest.approx(fde, rel=1e-4)
    # the aggregated stress constraint of the shell drivers (shell_pde.py:297-313): value and total derivative
    Sref = lambda hh: so.pnorm_stress(V0, solve_ref(hh), hh, E_ROOF, NU_ROOF, m=2e-6, rho=4.0)
    assert sim['pnorm_stress'][0] == pytest.approx(Sref(h), rel=1e-7)
    vm = np.asarray(sim['von_Mises_stress'])
    vref = so.project_von_mises(V0, wref, h, E_ROOF, NU_ROOF, 1.0)
    assert vm.shape == (V0.n_vert,) and np.abs(vm - vref).max() <= 1e-7 * np.abs(vref).max()
    assert np.array_equal(pde.projected_von_Mises_stress(pde.von_Mises_stress(w_fn, h_fn, E_ROOF, NU_ROOF)).vec.get(), vm)
    gs = np.asarray(sim.compute_totals('pnorm_stress', 'thickness'))
    assert gs @ dh == pytest.approx((Sref(h + 1e-2 * dh) - Sref(h - 1e-2 * dh)) / 2e-2, rel=1e-4)


def test_lattice_preconditioner(ctx):
    """opts->pc = 1: same solution as Jacobi-PCG in a fraction of the iterations (8 x 8 / 16 x 16 / 32 x 32 roof: 349 / 267 /
    166 against 1137 / 3169 / 7490; with the diagonal levels only, the first version, 16 x 16 took ~980): the count does
    not grow with the mesh where Jacobi's doubles."""
    from femo_amd.fea.shell import ShellProblem
    res = {}
    for n in (8, 16):
        pts, conn = so.scordelis_lo_mesh(n, n)
        V0 = so.ShellSpace(pts, conn)
        for pc in ("jacobi", "lattice"):
            prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=roof_fixed(V0), ctx=ctx, pc=pc)
            prob.set_thickness(H_ROOF)
            prob.set_load([0.0, 0.0, FZ])
            w = prob.solve(rtol=1e-10)
            res[(n, pc)] = (prob.last_info.iterations, w)
    for n in (8, 16):
        assert rel(res[(n, "lattice")][1], res[(n, "jacobi")][1]) <= 1e-7
        assert res[(n, "lattice")][0] < res[(n, "jacobi")][0]
    assert res[(16, "lattice")][0] < 0.4 * res[(16, "jacobi")][0]           # 267 against 3169
    assert res[(16, "lattice")][0] < 1.4 * res[(8, "lattice")][0]           # 349 -> 267; Jacobi: 1137 -> 3169
    assert res[(16, "jacobi")][0] > 2.0 * res[(8, "jacobi")][0]


def test_coarse_solve(ctx):
    """Exact coarse solve of the lattice preconditioner: the dense Galerkin operator formed on the device is
    P_c^T K_ff P_c of the oracle's stiffness, its inverse is one, the solution is unchanged and the iteration count
    drops by more than half (983 -> 357 on the 16 x 16 roof with the NumPy restatement)."""
    import scipy.sparse as sp
    from femo_amd.fea.shell import ShellProblem, ShellSpace, coarse_solve_plan, lattice_pc
    n = 16
    pts, conn = so.scordelis_lo_mesh(n, n)
    V0 = so.ShellSpace(pts, conn)
    fixed = roof_fixed(V0)
    res = {}
    for coarse in (0, 3200):
        prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=fixed, ctx=ctx, pc="lattice")
        prob.dev.enable_lattice_pc(coarse_unknowns=coarse, hermite=False)      # the trilinear hierarchy (its Hermite-type successor: test_gpu_shell_hermite.py)
        prob.set_thickness(H_ROOF)
        prob.set_load([0.0, 0.0, FZ])
        w = prob.solve(rtol=1e-10)
        res[coarse] = (prob.last_info.iterations, w)
    assert rel(res[3200][1], res[0][1]) <= 1e-7
    assert res[3200][0] < 0.5 * res[0][0]
    # the matrix itself
    L = lattice_pc(ShellSpace(pts, conn))
    plan = coarse_solve_plan(L)
    c, off = plan["level"], L["level_offsets"]
    assert prob.dev.coarse_level == c
    K = so.assemble(V0, so.element_stiffness(V0, np.full(V0.n_vert, H_ROOF), E_ROOF, NU_ROOF)).tocsr()
    free = np.ones(V0.n_dof, dtype=bool); free[fixed] = False
    D = sp.diags(free.astype(float))
    rows = np.repeat(np.arange(V0.n_dof), 8)
    P = sp.csr_matrix((L["ell_w"][:, 8 * c:8 * c + 8].ravel(), (rows, L["ell_idx"][:, 8 * c:8 * c + 8].ravel())),
                      shape=(V0.n_dof, L["n_lat"]))[:, 6 * off[c]:6 * off[c + 1]]
    A_ref = (P.T @ (D @ K @ D) @ P).toarray()
    mask = (~free).astype(np.uint8)
    A = prob.dev.coarse_matrix(prob.vals, mask)
    assert A.shape == A_ref.shape
    assert np.abs(A - A_ref).max() <= 1e-11 * np.abs(A_ref).max()
    W = prob.dev.coarse_matrix(prob.vals, mask, inverse=True)          # L^-T above, L^-1 below the diagonal
    d = np.diag(A_ref).copy()
    A_fix = A_ref + np.diag(np.where(d > 0, 1e-13 * d, 1.0))
    assert np.array_equal(np.triu(W), np.tril(W).T)
    Li = np.tril(W)
    assert np.abs(Li @ A_fix @ Li.T - np.eye(A.shape[0])).max() <= 1e-6


def test_preconditioner_variants_agree(ctx, monkeypatch):
    """The parts of the preconditioner can be switched off one by one (environment switches read at solve time): the
    solution does not change, the iteration count does -- every part earns its place on the 32 x 32 roof."""
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(32, 32)
    V0 = so.ShellSpace(pts, conn)
    fixed = roof_fixed(V0)
    res = {}
    for name, env in (("full", {}), ("no node blocks", {"FEMO_SHELL_NO_BLOCKS": "1"}),
                      ("no point blocks", {"FEMO_SHELL_NO_POINT_BLOCKS": "1"}), ("csr view", {"FEMO_SHELL_NO_BSELL": "1"})):
        for k in ("FEMO_SHELL_NO_BLOCKS", "FEMO_SHELL_NO_POINT_BLOCKS", "FEMO_SHELL_NO_BSELL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=fixed, ctx=ctx, pc="lattice")
        prob.set_thickness(H_ROOF)
        prob.set_load([0.0, 0.0, FZ])
        w = prob.solve(rtol=1e-10)
        res[name] = (prob.last_info.iterations, w)
    for name in res:
        assert rel(res[name][1], res["full"][1]) <= 1e-7, name
    assert res["full"][0] < res["no node blocks"][0]
    assert res["full"][0] < res["no point blocks"][0]
    assert abs(res["csr view"][0] - res["full"][0]) <= 10         # same operator, another summation order


def test_irregular_surface_mesh(ctx):
    """Jittered vertices (moved along the cylinder), flipped and rotated cell numbering: kernels against the oracle,
    the lattice-preconditioned solve against the direct one; and a flat plate, whose bounding box is degenerate."""
    from femo_amd.fea.shell import ShellProblem
    rng = np.random.default_rng(13)
    pts, conn = so.scordelis_lo_mesh(10, 10)
    X = pts[:, 0] + rng.uniform(-0.6, 0.6, len(pts)) * (pts[:, 0] > 1e-6) * (pts[:, 0] < L - 1e-6)
    phi = np.arcsin(pts[:, 1] / 25.0)
    inner = (phi > 1e-9) & (phi < phi.max() - 1e-9)
    phi = phi + rng.uniform(-0.015, 0.015, len(pts)) * inner
    pts = np.stack([X, 25.0 * np.sin(phi), 25.0 * np.cos(phi)], axis=1)
    conn = conn.copy()
    flip = rng.random(len(conn)) < 0.5
    conn[flip] = conn[flip][:, [0, 2, 1]]
    conn = np.stack([np.roll(c, r) for c, r in zip(conn, rng.integers(0, 3, len(conn)))])
    V0 = so.ShellSpace(pts, conn)
    fixed = roof_fixed(V0)
    prob = ShellProblem(pts, conn, E_ROOF, 0.25, fixed_dofs=fixed, ctx=ctx)
    h = H_ROOF * (1.0 + 0.3 * rng.random(V0.n_vert))
    f = rng.standard_normal((V0.n_vert, 3)) * 50.0
    prob.set_thickness(h); prob.set_load(f)
    Kref = so.assemble(V0, so.element_stiffness(V0, h, E_ROOF, 0.25))
    rowptr, cols, _ = prob.space.pattern()
    K = sp.csr_matrix((np.array(prob._stiffness().get()), cols, rowptr), shape=Kref.shape)
    assert abs(K - Kref).max() <= 1e-12 * abs(Kref).max()
    wref = so.solve(Kref, so.load_vector(V0, f), fixed)
    assert rel(prob.solve(rtol=1e-11), wref) <= 1e-7
    # flat plate in the plane z = 0
    ppts, pconn = so.plate_mesh(12)
    P0 = so.ShellSpace(ppts, pconn)
    edge = np.nonzero(np.isclose(P0.unode_x[:, 0], 0) | np.isclose(P0.unode_x[:, 0], 1))[0]
    pfixed = np.unique(np.concatenate([P0.u_dof(edge, k) for k in range(3)] + [P0.theta_dof(np.nonzero(np.isclose(P0.x[:, 0], 0))[0], k) for k in range(3)]))
    plate = ShellProblem(ppts, pconn, 1.0e7, 0.3, fixed_dofs=pfixed, ctx=ctx)
    plate.set_thickness(0.02); plate.set_load([0.0, 0.0, -1.0])
    Kp = so.assemble(P0, so.element_stiffness(P0, np.full(P0.n_vert, 0.02), 1.0e7, 0.3))
    wp = so.solve(Kp, so.load_vector(P0, np.tile([0.0, 0.0, -1.0], (P0.n_vert, 1))), pfixed)
    assert rel(plate.solve(rtol=1e-11), wp) <= 1e-6


def test_thickness_optimisation_of_the_roof(ctx):
    """The use the shell path is built for (run_shape_opt_roof.py:163-213: compliance objective, volume constraint,
    thickness design variable): a few SLSQP iterations driven by the GPU operators' values and adjoint gradients
    lower the compliance at constant mass."""
    import scipy.optimize as sopt
    from femo_amd.fea.shell import ShellProblem
    pts, conn = so.scordelis_lo_mesh(8, 8)
    V0 = so.ShellSpace(pts, conn)
    prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=roof_fixed(V0), ctx=ctx)
    prob.set_load([0.0, 0.0, FZ])
    h0 = np.full(V0.n_vert, H_ROOF)
    prob.set_thickness(h0)
    m0 = prob.mass(1.0)
    cache = {}

    def evaluate(h):
        key = h.tobytes()
        if key not in cache:
            cache.clear()
            prob.set_thickness(h)
            J, g, _ = prob.compliance_gradient()
            M, gm = prob.mass(1.0, grad=True)
            cache[key] = (J, g, M, gm)
        return cache[key]

    J0 = evaluate(h0)[0]
    res = sopt.minimize(lambda h: evaluate(h)[0] / J0, h0, jac=lambda h: evaluate(h)[1] / J0, method="SLSQP",
                        bounds=[(0.4 * H_ROOF, 2.5 * H_ROOF)] * V0.n_vert,
                        constraints=[dict(type="ineq", fun=lambda h: (m0 - evaluate(h)[2]) / m0, jac=lambda h: -evaluate(h)[3] / m0)],
                        options=dict(maxiter=12, ftol=1e-9))
    J1, _, M1, _ = evaluate(res.x)
    assert M1 <= m0 * (1 + 1e-6)
    assert J1 < 0.8 * J0                                   # material moves towards the free edge and the diaphragm
    assert res.x.max() > 1.2 * H_ROOF and res.x.min() < 0.9 * H_ROOF
