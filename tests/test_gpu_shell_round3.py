"""Round 3 of the shell path: the boundary conditions as the reference's driver configures them -- penalty terms on tagged
exterior / interior facets (`shell_pde.py:34,59-61,246-253`) --, `kinetic_residual` (`:255-256`), the regularisation options of
`compliance` and its `dxx` subset (`:262-285`), pnorm_stress(regularization=True) (`:307-309`), and exact (not
finite-difference) checks of the thickness partials.  Every kernel against oracle/shell_oracle.py."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import shell_oracle as so

pytestmark = pytest.mark.gpu

E_Y, NU = 7.0e7, 0.3


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


@pytest.fixture(scope="module")
def wing(ctx):
    """A gently curved cantilever panel, clamped (by penalty) over a strip at its root."""
    from femo_amd.engine import Vec
    from femo_amd.fea.shell import DeviceShell, ShellSpace
    n = 10
    xs = np.linspace(0.0, 2.0, 2 * n + 1)
    ys = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), 0.15 * np.sin(np.pi * Y.ravel()) * (1.0 + 0.2 * X.ravel())], axis=1)
    idx = np.arange(pts.shape[0]).reshape(2 * n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    conn = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    V = so.ShellSpace(pts, conn)
    S = ShellSpace(pts, conn)
    dev = DeviceShell(ctx, S)
    rng = np.random.default_rng(7)
    h = 0.02 * (1.0 + 0.3 * rng.random(V.n_vert))
    return dict(V=V, S=S, dev=dev, h=h, rng=rng, ctx=ctx, Vec=Vec, pts=pts, conn=conn)


def test_penalty_matrix_and_residual(wing):
    V, S, dev, h, rng, ctx, Vec = (wing[k] for k in ("V", "S", "dev", "h", "rng", "ctx", "Vec"))
    root = lambda x: x[0] <= 0.2 + 1e-9                       # a clamped REGION: boundary and interior facets (ds and dS)
    ext, inte = S.tagged_edges(root)
    e0, i0 = so.tagged_edges(V, root)
    assert np.array_equal(ext, e0) and np.array_equal(inte, i0) and len(ext) > 0 and len(inte) > 0
    edges = np.concatenate([ext, inte])
    rowptr, cols, _ = S.pattern()
    hv = Vec(ctx, V.n_vert).set(h)
    for beta in (1e6, 1e15):
        dev.set_penalty(edges, beta)
        vals = Vec(ctx, dev.nnz)
        dev.assemble(E_Y, NU, hv, vals)
        k0 = np.array(vals.get())
        dev.penalty_add(vals)
        Kp = sp.csr_matrix((np.array(vals.get()) - k0, cols, rowptr), shape=(S.n_dof, S.n_dof))
        Kref = so.penalty_matrix(V, ext, inte, beta)
        assert abs(Kp - Kref).max() <= 1e-12 * abs(Kref).max()
        assert abs(Kp - Kp.T).max() <= 1e-12 * abs(Kref).max()
        w, g = rng.standard_normal(S.n_dof), rng.standard_normal(S.n_dof)
        y = dev.penalty_apply(Vec(ctx, S.n_dof).set(w), Vec(ctx, S.n_dof), g=Vec(ctx, S.n_dof).set(g))
        assert rel(y.get(), Kref @ (w - g)) <= 1e-13
        y2 = Vec(ctx, S.n_dof).set(w)
        dev.penalty_apply(Vec(ctx, S.n_dof).set(w), y2, accumulate=True)          # homogeneous data, accumulate
        assert rel(y2.get(), w + Kref @ w) <= 1e-13
    dev.set_penalty(np.zeros(0, np.int64), 1.0)                                   # none: both calls are no-ops
    vals = Vec(ctx, dev.nnz).fill(1.0)
    dev.penalty_add(vals)
    assert np.all(np.array(vals.get()) == 1.0)
    with pytest.raises(Exception):
        S.positions([0], [S.n_dof - 1])                                           # not a coupling of the pattern


def test_inertia_regularisation_and_weighted_compliance(wing):
    V, S, dev, h, rng, ctx, Vec = (wing[k] for k in ("V", "S", "dev", "h", "rng", "ctx", "Vec"))
    hv = Vec(ctx, V.n_vert).set(h)
    a, lam = rng.standard_normal(S.n_dof), rng.standard_normal(S.n_dof)
    av, lv = Vec(ctx, S.n_dof).set(a), Vec(ctx, S.n_dof).set(lam)
    rho = 2710.0
    y = dev.inertia_apply(rho, hv, av, Vec(ctx, S.n_dof))
    assert rel(y.get(), so.inertia_apply(V, h, rho, a)) <= 1e-13
    # lam^T dM/dh a: M is cubic in h, Richardson-extrapolated central differences of the oracle are exact up to round-off
    g = np.array(dev.inertia_dh(rho, hv, lv, av, Vec(ctx, V.n_vert)).get())
    dh = 0.1 * h * rng.standard_normal(V.n_vert)
    form = lambda hh: float(lam @ so.inertia_apply(V, hh, rho, a))
    d1, d2 = (form(h + dh) - form(h - dh)) / 2.0, (form(h + 2 * dh) - form(h - 2 * dh)) / 4.0
    assert g @ dh == pytest.approx((4 * d1 - d2) / 3, rel=1e-10)
    for kind in ("H1", "L2H1", "L2"):
        gv = Vec(ctx, V.n_vert)
        val = dev.regularization(kind, hv, grad=gv)
        vref, gref = so.regularization(V, h, kind, grad=True)
        assert val == pytest.approx(vref, rel=1e-13) and rel(gv.get(), gref) <= 1e-12
    cells = np.nonzero(V.x[V.conn][:, :, 0].min(axis=1) >= 1.5)[0]               # a tip region, like dx_2(10)
    cw = np.zeros(S.n_cell); cw[cells] = 1.0
    w = rng.standard_normal(S.n_dof)
    wv, gw = Vec(ctx, S.n_dof).set(w), Vec(ctx, S.n_dof)
    J = dev.compliance_dx(wv, Vec(ctx, S.n_cell).set(cw), grad=gw)
    assert J == pytest.approx(so.compliance(V, w, cells), rel=1e-13)
    assert rel(gw.get(), so.compliance_du(V, w, cells)) <= 1e-13
    assert dev.compliance_dx(wv, None) == pytest.approx(so.compliance(V, w), rel=1e-13)
    # int c h^p with the degree-4 rule
    _, _, _, area, _ = V.frames()
    lam6, w6 = so.QUAD_INPLANE
    hq = h[V.conn] @ lam6.T
    for p in (1.0, 3.0, 100.0):
        gp = Vec(ctx, V.n_vert)
        val = dev.hpower(0.5e3, p, hv, grad=gp)
        ref = 0.5e3 * float((area[:, None] * w6[None, :] * hq ** p).sum())
        assert val == pytest.approx(ref, rel=1e-12)
        gref = np.zeros(V.n_vert)
        np.add.at(gref, V.conn.ravel(), (0.5e3 * p * np.einsum("c,q,cq,qb->cb", area, w6, hq ** (p - 1.0), lam6)).ravel())
        assert rel(gp.get(), gref) <= 1e-12


def test_thickness_partial_of_the_stiffness_is_exact(wing):
    """(dR/dh)^T lambda against the oracle's exact derivative (round 2 compared with finite differences at 1e-6)."""
    V, S, dev, h, rng, ctx, Vec = (wing[k] for k in ("V", "S", "dev", "h", "rng", "ctx", "Vec"))
    v, w = rng.standard_normal(S.n_dof), rng.standard_normal(S.n_dof)
    out = dev.dform_dh(E_Y, NU, Vec(ctx, V.n_vert).set(h), Vec(ctx, S.n_dof).set(v), Vec(ctx, S.n_dof).set(w), out=Vec(ctx, V.n_vert))
    assert rel(out.get(), so.dform_dh(V, h, E_Y, NU, v, w)) <= 1e-12


def test_forward_products_with_the_thickness_partials(wing):
    """Round 5 (missing #4): compute_jacvec_product(mode='fwd') with dR/dh and dM/dh
    (state_model.py:176-188).  The forward product is the exact transpose of the reverse one, <v, (dK/dh [dh]) w> =
    <dh, dform_dh(v, w)>, and a directional derivative of the residual: K is cubic in h, so a Richardson-extrapolated
    central difference of the oracle's K(h) w is exact up to round-off; the same for M(h) a.  Through the form layer too
    (`_ShellPartial.mult`, `_InertiaOperator.mult`), where the reference's fwd branch would call them."""
    V, S, dev, h, rng, ctx, Vec = (wing[k] for k in ("V", "S", "dev", "h", "rng", "ctx", "Vec"))
    v, w = rng.standard_normal(S.n_dof), rng.standard_normal(S.n_dof)
    dh = 0.1 * h * rng.standard_normal(V.n_vert)
    hv, dhv, wv = Vec(ctx, V.n_vert).set(h), Vec(ctx, V.n_vert).set(dh), Vec(ctx, S.n_dof).set(w)
    y = np.array(dev.dform_dh_fwd(E_Y, NU, hv, dhv, wv, Vec(ctx, S.n_dof)).get())
    out = np.array(dev.dform_dh(E_Y, NU, hv, Vec(ctx, S.n_dof).set(v), wv, out=Vec(ctx, V.n_vert)).get())
    assert v @ y == pytest.approx(dh @ out, rel=1e-12)
    Kw = lambda hh: so.assemble(V, so.element_stiffness(V, hh, E_Y, NU)).tocsr() @ w
    d1, d2 = (Kw(h + dh) - Kw(h - dh)) / 2.0, (Kw(h + 2 * dh) - Kw(h - 2 * dh)) / 4.0
    assert rel(y, (4 * d1 - d2) / 3) <= 1e-10
    y2 = Vec(ctx, S.n_dof).set(w)                                               # accumulate
    dev.dform_dh_fwd(E_Y, NU, hv, dhv, wv, y2, accumulate=True)
    assert rel(y2.get(), w + y) <= 1e-13
    # inertia
    rho = 2710.0
    a, lam = rng.standard_normal(S.n_dof), rng.standard_normal(S.n_dof)
    av = Vec(ctx, S.n_dof).set(a)
    ym = np.array(dev.inertia_dh_fwd(rho, hv, dhv, av, Vec(ctx, S.n_dof)).get())
    gm = np.array(dev.inertia_dh(rho, hv, Vec(ctx, S.n_dof).set(lam), av, Vec(ctx, V.n_vert)).get())
    assert lam @ ym == pytest.approx(dh @ gm, rel=1e-12)
    Ma = lambda hh: so.inertia_apply(V, hh, rho, a)
    d1, d2 = (Ma(h + dh) - Ma(h - dh)) / 2.0, (Ma(h + 2 * dh) - Ma(h - 2 * dh)) / 4.0
    assert rel(ym, (4 * d1 - d2) / 3) <= 1e-10


def _penalty_problem(wing, beta):
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA
    from femo_amd.fea.function import Function
    from femo_amd.fea.shell_forms import ShellMesh, ShellPDE, createCustomMeasure
    utils_hip.set_context(wing["ctx"])
    mesh = ShellMesh(wing["pts"], wing["conn"])
    pde = ShellPDE(mesh)
    root = lambda x: x[0] <= 0.2 + 1e-9
    tipr = lambda x: x[0] >= 1.5 - 1e-9
    ds_1 = createCustomMeasure(mesh, 1, root, measure='ds', tag=100)
    dS_1 = createCustomMeasure(mesh, 1, root, measure='dS', tag=100)
    dx_2 = createCustomMeasure(mesh, 2, tipr, measure='dx', tag=10)
    fea = FEA(mesh)
    fea.REPORT = False
    fea.linear_problem = True
    h_fn, f_fn, w_fn, g_fn = Function(pde.VT), Function(pde.VF), Function(pde.W), Function(pde.W)
    g_fn.vector.set(0.0)
    res = pde.pdeRes(h_fn, w_fn, f_fn, E_Y, NU, penalty=True, dss=ds_1(100), dSS=dS_1(100), g=g_fn, beta=beta)   # shell_pde.py:59-61
    fea.add_input('thickness', h_fn, init_val=0.02)
    fea.add_input('F_solid', f_fn, init_val=0.0)
    fea.add_state(name='disp_solid', function=w_fn, residual_form=res, arguments=['thickness', 'F_solid'])
    fea.add_output(name='compliance', type='scalar', form=pde.compliance(w_fn, h_fn, dx_2(10)), arguments=['disp_solid', 'thickness'])
    fea.add_output(name='compliance_reg', type='scalar', form=pde.compliance(w_fn, h_fn, dx_2(10), regularization='L2H1'),
                   arguments=['disp_solid', 'thickness'])
    fea.add_output(name='elastic_energy', type='scalar', form=pde.elastic_energy(w_fn, h_fn, E_Y), arguments=['disp_solid', 'thickness'])
    model = FEAModel(fea=[fea])
    f = np.tile([0.0, 0.0, -50.0], (mesh.n_vert, 1)).ravel()
    model.create_input('thickness', shape=mesh.n_vert, val=wing["h"])
    model.create_input('F_solid', shape=3 * mesh.n_vert, val=f)
    return Simulator(model), pde, fea, dict(root=root, tipr=tipr, f=f, res=res, g_fn=g_fn, h_fn=h_fn, w_fn=w_fn, f_fn=f_fn)


def test_operator_cycle_with_penalty_boundary_conditions(wing):
    """`ShellPDE.pdeRes(penalty=True, dss, dSS, g)` behind FEA / StateOperation / OutputOperation, as shell_pde.py:20-120
    registers it: state, outputs on the tagged `dxx` cells and the adjoint thickness sensitivity against the oracle's direct
    solve of (K + K_pen) and its EXACT adjoint gradient; the strong-BC limit."""
    V, h = wing["V"], wing["h"]
    beta = 1e15
    sim, pde, fea, X = _penalty_problem(wing, beta)
    sim.run()
    ext, inte = so.tagged_edges(V, X["root"])
    K = so.assemble(V, so.element_stiffness(V, h, E_Y, NU))
    Kp = so.penalty_matrix(V, ext, inte, beta)
    F = so.load_vector(V, X["f"].reshape(-1, 3))
    lu = spla.splu((K + Kp).tocsc())
    wref = lu.solve(F)
    w = np.asarray(sim['disp_solid'])
    assert rel(w, wref) <= 1e-8
    cells = np.nonzero(X["tipr"](V.x.T)[V.conn].all(axis=1))[0]
    assert sim['compliance'][0] == pytest.approx(so.compliance(V, wref, cells), rel=1e-8)
    assert sim['compliance_reg'][0] == pytest.approx(so.compliance(V, wref, cells) + so.regularization(V, h, 'L2H1'), rel=1e-9)
    # elastic_energy(w, h, E) takes the Poisson ratio of the elastic model built by pdeRes, no penalty energy in it
    assert sim['elastic_energy'][0] == pytest.approx(sum(so.energy_parts(V, wref, h, E_Y, NU).values()), rel=1e-8)
    # adjoint total: dJ/dh = -lam^T dK/dh w (the penalty terms do not depend on h), (K + K_pen) lam = dJ/dw
    lam = lu.solve(so.compliance_du(V, wref, cells))
    gref = -so.dform_dh(V, h, E_Y, NU, lam, wref)
    g = np.asarray(sim.compute_totals('compliance', 'thickness'))
    assert rel(g, gref) <= 1e-7
    _, greg = so.regularization(V, h, 'L2H1', grad=True)
    assert rel(np.asarray(sim.compute_totals('compliance_reg', 'thickness')), gref + greg) <= 1e-7
    # evaluate_residuals: (K + K_pen) w - F - K_pen g with inhomogeneous data
    gv = np.random.default_rng(3).standard_normal(V.n_dof) * 1e-3
    X["g_fn"].vector[:] = gv
    X["w_fn"].vector[:] = wref
    from femo_amd.fea.utils_hip import assembleVector
    r = assembleVector(X["res"])
    rref = (K + Kp) @ wref - F - Kp @ gv
    assert np.abs(r - rref).max() <= 1e-9 * np.abs(Kp @ gv).max()
    X["g_fn"].vector.set(0.0)
    # forward mode of the operator (round 5): d_residuals += dR/dh . d_h + dR/df . d_f through compute_jacvec_product
    op = [o for _, o in sim.ops if hasattr(o, 'apply_inverse_jacobian')][0]
    ins, outs = {'thickness': sim.values['thickness'], 'F_solid': sim.values['F_solid']}, {'disp_solid': sim.values['disp_solid']}
    op.compute_derivatives(ins, outs, {})
    rng5 = np.random.default_rng(5)
    dh, df = 0.1 * h * rng5.standard_normal(V.n_vert), rng5.standard_normal(3 * V.n_vert)
    d_res = {'disp_solid': np.zeros(V.n_dof)}
    op.compute_jacvec_product(ins, outs, {'thickness': dh, 'F_solid': df}, {}, d_res, 'fwd')
    Kw = lambda hh: so.assemble(V, so.element_stiffness(V, hh, E_Y, NU)).tocsr() @ w
    d1, d2 = (Kw(h + dh) - Kw(h - dh)) / 2.0, (Kw(h + 2 * dh) - Kw(h - 2 * dh)) / 4.0
    ref_fwd = (4 * d1 - d2) / 3 - so.load_vector(V, df.reshape(-1, 3))
    assert rel(np.asarray(d_res['disp_solid']), ref_fwd) <= 1e-9
    # strong-BC limit: clamp every dof on the tagged edges strongly
    un = np.unique(np.concatenate([V.edge_vertices[np.concatenate([ext, inte])].ravel(), V.n_vert + np.concatenate([ext, inte])]))
    vn = un[un < V.n_vert]
    fixed = np.concatenate([V.u_dof(un, k) for k in range(3)] + [V.theta_dof(vn, k) for k in range(3)])
    ws = so.solve(K, F, fixed)
    assert rel(w, ws) <= 1e-6


def test_penalty_refused_without_measures_and_kinetic_residual(wing):
    from femo_amd.fea import utils_hip
    from femo_amd.fea.function import Function
    from femo_amd.fea.shell_forms import ShellMesh, ShellPDE, createCustomMeasure
    from femo_amd.fea.utils_hip import assembleVector, computeMatVecProductBwd, assembleMatrix, computePartials
    utils_hip.set_context(wing["ctx"])
    V, h, rng = wing["V"], wing["h"], wing["rng"]
    mesh = ShellMesh(wing["pts"], wing["conn"])
    pde = ShellPDE(mesh)
    h_fn, f_fn, w_fn, a_fn, lam_fn = Function(pde.VT), Function(pde.VF), Function(pde.W), Function(pde.W), Function(pde.W)
    with pytest.raises(ValueError):
        pde.pdeRes(h_fn, w_fn, f_fn, E_Y, NU, penalty=True)
    with pytest.raises(TypeError):
        pde.pdeRes(h_fn, w_fn, f_fn, E_Y, NU, penalty=True, dss=createCustomMeasure(mesh, 2, lambda x: x[0] < 1, measure='dx'))
    h_fn.vector[:] = h
    a = rng.standard_normal(V.n_dof)
    a_fn.vector[:] = a
    kin = pde.kinetic_residual(2710.0, h_fn, a_fn)
    assert rel(assembleVector(kin), so.inertia_apply(V, h, 2710.0, a)) <= 1e-13
    lam = rng.standard_normal(V.n_dof)
    lam_fn.vector[:] = lam
    dMa = assembleMatrix(computePartials(kin, a_fn))
    assert rel(computeMatVecProductBwd(dMa, lam_fn), so.inertia_apply(V, h, 2710.0, lam)) <= 1e-13
    dMh = assembleMatrix(computePartials(kin, h_fn))
    g = computeMatVecProductBwd(dMh, lam_fn)
    assert g.shape == (V.n_vert,)
    # elastic_residual: K(h) w alone
    w = rng.standard_normal(V.n_dof) * 1e-3
    w_fn.vector[:] = w
    f_fn.vector.set(3.0)
    er = pde.elastic_residual(h_fn, w_fn, f_fn, E_Y, NU)
    Kw = so.assemble(V, so.element_stiffness(V, h, E_Y, NU)) @ w
    assert rel(assembleVector(er), Kw) <= 1e-11
    # pnorm_stress(regularization=True) adds 1/alpha int 0.5e3 h^rho
    ps = pde.pnorm_stress(w_fn, h_fn, E_Y, NU, m=2e-6, rho=4, regularization=True)
    from femo_amd.fea.utils_hip import assembleScalar
    _, _, _, area, _ = V.frames()
    lam6, w6 = so.QUAD_INPLANE
    extra = 0.5e3 * float((area[:, None] * w6[None, :] * (h[V.conn] @ lam6.T) ** 4).sum()) / area.sum()
    assert assembleScalar(ps) == pytest.approx(so.pnorm_stress(V, w, h, E_Y, NU, m=2e-6, rho=4.0) + extra, rel=1e-10)
