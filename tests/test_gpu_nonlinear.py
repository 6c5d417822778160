"""GPU parity of the nonlinear Poisson + symmetric Nitsche form (BASELINE config 5,
examples/nonlinear_poisson_opt) against the CPU oracle."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - b).max() / np.abs(b).max()


@pytest.mark.parametrize("d,n,jit,facets", [(2, 9, 0.2, True), (2, 40, 0.0, True), (3, 5, 0.2, True), (3, 6, 0.0, False)])
def test_nl_kernels_match_oracle(ctx, d, n, jit, facets):
    from femo_amd import engine as E
    m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    rng = np.random.default_rng(5)
    u, f = 0.7 * rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell)
    uex = fo.u_exact_nl(m.x)
    bm = fo.boundary_facets(m) if facets else np.zeros(m.n_cell, np.uint8)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    if facets:
        dm.set_boundary_facets(bm)
    U, F, UEX = E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).set(f), E.Vec(ctx, m.n_vert).set(uex)
    beta = 10.0
    R = E.Vec(ctx, m.n_vert)
    E.assemble_residual(dm, 1, [beta], U, F, R, aux=UEX)
    r_ref = fo.nl_residual(m, u, f, uex, bm, beta)
    assert _rel(R.get(), r_ref) < 1e-12
    J = E.Mat(dm)
    E.assemble_jacobian(dm, 1, [beta], U, F, None, J, aux=UEX)
    J_ref = fo.nl_jacobian(m, u, bm, beta)
    Jg = J.to_scipy()
    assert np.array_equal(Jg.indices, J_ref.indices) and _rel(Jg.data, J_ref.data) < 1e-12
    # fused: Jacobian + Newton right-hand side (no Dirichlet set: rhs = residual)
    J2, B = E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 1, [beta], U, F, None, J2, None, B, aux=UEX)
    assert _rel(B.get(), r_ref) < 1e-12 and np.array_equal(J2.to_scipy().data, Jg.data)


@pytest.mark.parametrize("d,n,device", [(2, 24, False), (2, 32, True), (3, 8, True)])
def test_nl_cycle_matches_oracle(ctx, d, n, device):
    """run_nonlinear_poisson_opt.py:147-232 on the mirror: DG0 f = 0.1, SNES, weak BCs (sym Nitsche)."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    from femo_amd.fea.nonlinear_poisson import ALPHA_1, outputForm, pdeRes
    utils_hip.set_context(ctx)
    mesh = createUnitSquareMesh(n) if d == 2 else createUnitCubeMesh(n)
    om = fo.unit_square_mesh(n) if d == 2 else fo.unit_cube_mesh(n)
    fea = FEA(mesh)
    fea.REPORT = False
    Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
    f_fn, u_fn = Function(Vf), Function(Vu)
    v = TestFunction(Vu)
    u_ex = Function(Vu)
    u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]) * (np.sin(np.pi * x[2]) if d == 3 else 1.0))
    residual_form = pdeRes(u_fn, v, f_fn, u_exact=u_ex, weak_bc=True, sym=True)
    fea.add_input('f', f_fn)
    fea.add_state(name='u', function=u_fn, residual_form=residual_form, arguments=['f'])
    fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
    fea.PDE_SOLVER = 'SNES'
    model = FEAModel(fea=[fea])
    model.create_input('f', shape=fea.inputs_dict['f']['shape'], val=0.1)
    sim = Simulator(model, device=device)
    sim.run()
    ref = fo.nl_reference_cycle(om, 0.1 * np.ones(om.n_cell), fo.u_exact_nl(om.x), fo.boundary_facets(om), ALPHA_1)
    assert _rel(sim['u'], ref['u']) < 1e-10
    assert abs(sim['l2_functional'][0] - ref['J'][0]) < 1e-10 * abs(ref['J'][0])
    g = np.asarray(sim.compute_totals('l2_functional', 'f'))
    assert _rel(g, ref['grad']) < 1e-10
    # no Dirichlet rows: the adjoint total is the exact reduced gradient -> finite differences agree
    chk = sim.check_totals('l2_functional', 'f', step=1e-5, n_dir=2)
    assert max(chk['rel_error']) < 1e-6


@pytest.mark.parametrize("d,n", [(2, 40), (3, 14)])
def test_bicgstab_matches_lu(ctx, d, n):
    """Non-symmetric operator (unsymmetric Nitsche, sgn = -1, no penalty): forward and transposed solves."""
    from femo_amd import engine as E
    import scipy.sparse.linalg as spla
    m = fo.unit_square_mesh(n, 0.2) if d == 2 else fo.unit_cube_mesh(n, 0.2)
    rng = np.random.default_rng(9)
    u, f, uex = 0.3 * rng.standard_normal(m.n_vert), rng.standard_normal(m.n_cell), fo.u_exact_nl(m.x)
    bm = fo.boundary_facets(m)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    dm.set_boundary_facets(bm)
    U, F, UEX = E.Vec(ctx, m.n_vert).set(u), E.Vec(ctx, m.n_cell).set(f), E.Vec(ctx, m.n_vert).set(uex)
    J, R = E.Mat(dm), E.Vec(ctx, m.n_vert)
    E.assemble_system(dm, 1, [0.0, -1.0], U, F, None, J, None, R, aux=UEX)
    Jo = fo.nl_jacobian(m, u, bm, 0.0, -1.0)
    assert _rel(J.to_scipy().data, Jo.data) < 1e-12 and abs(Jo - Jo.T).max() > 1e-3
    assert _rel(R.get(), fo.nl_residual(m, u, f, uex, bm, 0.0, -1.0)) < 1e-12
    b = rng.standard_normal(m.n_vert)
    B, X = E.Vec(ctx, m.n_vert).set(b), E.Vec(ctx, m.n_vert)
    lu = spla.splu(Jo.tocsc())
    info = J.solve_bicgstab(B, X, rtol=1e-13)
    assert info.converged == 1 and _rel(X.get(), lu.solve(b)) < 1e-9
    info = J.solve_bicgstab(B, X, transpose=True, rtol=1e-13)
    assert info.converged == 1 and _rel(X.get(), spla.splu(Jo.T.tocsc()).solve(b)) < 1e-9
    info = J.solve_bicgstab(B, X, transpose=True, rtol=1e-13, zero_guess=False)     # warm start
    assert info.converged == 1 and info.iterations <= 2
    # transposed SpMV through the explicit transpose permutation
    Y = E.Vec(ctx, m.n_vert)
    J.mult(U, Y, transpose=True)
    assert _rel(Y.get(), Jo.T @ u) < 1e-12


def test_nl_unsymmetric_cycle(ctx):
    """run_nonlinear_poisson_opt.py with sym=False: Newton with BiCGSTAB, adjoint with the true transpose."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
    from femo_amd.fea.mesh import createUnitSquareMesh
    from femo_amd.fea.nonlinear_poisson import ALPHA_1, outputForm, pdeRes
    utils_hip.set_context(ctx)
    n = 20
    mesh, om = createUnitSquareMesh(n), fo.unit_square_mesh(n)
    fea = FEA(mesh)
    fea.REPORT = False
    Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
    f_fn, u_fn, u_ex = Function(Vf), Function(Vu), Function(Vu)
    u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]))
    fea.add_input('f', f_fn)
    fea.add_state(name='u', function=u_fn, arguments=['f'],
                  residual_form=pdeRes(u_fn, TestFunction(Vu), f_fn, u_exact=u_ex, weak_bc=True, sym=False))
    fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
    fea.PDE_SOLVER = 'SNES'
    model = FEAModel(fea=[fea])
    model.create_input('f', shape=mesh.n_cell, val=0.1)
    sim = Simulator(model, device=True)
    sim.run()
    ref = fo.nl_reference_cycle(om, 0.1 * np.ones(om.n_cell), fo.u_exact_nl(om.x), fo.boundary_facets(om), ALPHA_1,
                                beta=0.0, sgn=-1.0)
    assert _rel(sim['u'], ref['u']) < 1e-9
    g = np.asarray(sim.compute_totals('l2_functional', 'f'))
    assert _rel(g, ref['grad']) < 1e-9
    chk = sim.check_totals('l2_functional', 'f', step=1e-5, n_dir=2)
    assert max(chk['rel_error']) < 1e-6


def test_nl_catalogue_limits(ctx):
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import Function, FunctionSpace
    from femo_amd.fea.mesh import createUnitSquareMesh
    from femo_amd.fea.nonlinear_poisson import pdeRes
    utils_hip.set_context(ctx)
    mesh = createUnitSquareMesh(4)
    u, f = Function(FunctionSpace(mesh, ('CG', 1))), Function(FunctionSpace(mesh, ('DG', 0)))
    assert pdeRes(u, None, f, u_exact=u, weak_bc=True, sym=False).is_symmetric is False
    assert pdeRes(u, None, f, u_exact=u, weak_bc=True, sym=True).is_symmetric is True
    with pytest.raises(ValueError):
        pdeRes(u, None, f, weak_bc=True, sym=True)                  # boundary data missing
