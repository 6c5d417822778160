"""Round-2 test gaps (VERDICT.md "Close test gaps"):
  * BASELINE config 5 at its full size (nonlinear Poisson + symmetric Nitsche, 2-D n = 2236, 5.0 M DOFs) through
    properties: SNES convergence, stationarity of the residual, the adjoint total against a directional finite
    difference of J, and the adjoint identity on the converged Jacobian;
  * the `custom_solve` hook of FEA.solve (fea_dolfinx.py:185-187; examples/em_motor_opt/run_motor_opt.py:131-166);
  * a randomly permuted vertex / cell numbering (no regular SELL slice, scattered gathers): kernels and the whole
    cycle against the oracle on the same permuted mesh, bitwise symmetry of K included;
  * the BPX-CG stopping rule (preconditioned residual): error against the direct solve at the stated tolerance,
    iteration counts that do not grow with the mesh, identity rows exact."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - b).max() / np.abs(b).max()


def _nl_problem(ctx, mesh, device):
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
    from femo_amd.fea.nonlinear_poisson import outputForm, pdeRes
    utils_hip.set_context(ctx)
    fea = FEA(mesh)
    fea.REPORT = False
    Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
    f_fn, u_fn = Function(Vf), Function(Vu)
    u_ex = Function(Vu)
    u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]))
    res = pdeRes(u_fn, TestFunction(Vu), f_fn, u_exact=u_ex, weak_bc=True, sym=True)
    fea.add_input('f', f_fn)
    fea.add_state(name='u', function=u_fn, residual_form=res, arguments=['f'])
    fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
    fea.PDE_SOLVER = 'SNES'
    model = FEAModel(fea=[fea])
    model.create_input('f', shape=fea.inputs_dict['f']['shape'], val=0.1)
    return Simulator(model, device=device), fea, res, u_fn, f_fn


def test_config5_full_size_properties(ctx):
    """5,004,169 DOFs / 9,999,392 cells on one GPU (BASELINE.json config 5 asks for 4; it fits one)."""
    from femo_amd import engine as E
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitSquareMesh
    n = 2236
    mesh = createUnitSquareMesh(n)
    assert mesh.n_vert == 5_004_169 and mesh.n_cell == 9_999_392
    sim, fea, res, u_fn, f_fn = _nl_problem(ctx, mesh, device=True)
    del utils_hip.LAST_KSP_INFO[:]
    sim.run()
    its = [i["iterations"] for i in utils_hip.LAST_KSP_INFO]
    assert 3 <= len(its) <= 8 and max(its) <= 60            # Newton steps (SNES, full step), BPX-CG its per step
    J0 = float(sim['l2_functional'][0])
    # stationarity: the assembled residual at the returned state is at round-off of its terms
    r = utils_hip.assembleVector(res)
    u = np.asarray(sim['u'])
    assert np.abs(r).max() < 1e-9 * max(1.0, np.abs(u).max())
    # adjoint total vs directional central finite difference of J (smooth direction)
    g = np.asarray(sim.compute_totals('l2_functional', 'f'))
    xc = mesh.centroids()
    d = np.cos(3.0 * xc[:, 0]) * np.sin(2.0 * xc[:, 1]) + 0.3
    f0 = 0.1 * np.ones(mesh.n_cell)
    vals = []
    for s in (+1.0, -1.0):
        sim['f'] = f0 + s * 1e-3 * d
        sim.run()
        vals.append(float(sim['l2_functional'][0]))
    fd = (vals[0] - vals[1]) / 2e-3
    an = float(g @ d)
    assert abs(an - fd) <= 5e-5 * abs(fd), (an, fd, J0)
    # adjoint identity on the Jacobian of the converged state: <A^-1 b, c> = <b, A^-T c>
    sim['f'] = f0
    sim.run()
    op = [o for _, o in sim.ops if hasattr(o, 'apply_inverse_jacobian')][0]
    op.compute_derivatives({'f': sim.values['f']}, {'u': sim.values['u']}, {})
    rng = np.random.default_rng(1)
    N = mesh.n_vert
    b, c = E.Vec(ctx, N).set(rng.standard_normal(N)), E.Vec(ctx, N).set(rng.standard_normal(N))
    xb, xc_ = E.Vec(ctx, N), E.Vec(ctx, N)
    op.A.mat.solve_cg(b, xb, rtol=1e-12, pc="bpx")
    op.A.mat.solve_cg(c, xc_, rtol=1e-12, pc="bpx")
    lhs, rhs = xb.dot(c), b.dot(xc_)
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)


def test_custom_solve_hook(ctx):
    """FEA.solve calls `custom_solve(res, func, bc, report)` instead of solveNonlinear when it is set and
    `initial_solve` is true (fea_dolfinx.py:185-187); an incremental (load-stepping) hook like the motor example's
    reaches the same state as the plain solve."""
    from femo_amd.fea.fea_hip import solveNonlinear
    from femo_amd.fea.mesh import createUnitSquareMesh
    mesh = createUnitSquareMesh(20)
    sim, fea, res, u_fn, f_fn = _nl_problem(ctx, mesh, device=False)
    sim['f'] = 0.3 * np.ones(mesh.n_cell)
    sim.run()
    u_plain = np.array(sim['u'])
    calls = []

    def incremental(res_, func, bc, report):
        """two load steps: f/2 first, then f, each from the previous state (run_motor_opt.py:131-166)"""
        calls.append((res_, func, bc, report))
        f_full = np.array(f_fn.vector.getArray())
        for frac in (0.5, 1.0):
            f_fn.vector[:] = frac * f_full
            solveNonlinear(res_, func, bc, 'SNES', report, False)

    fea.custom_solve = incremental
    u_fn.vector.set(1.0)
    sim['u'] = np.ones(mesh.n_vert)
    sim.run()
    assert len(calls) == 1 and calls[0][0] is res and calls[0][1] is u_fn and calls[0][3] is fea.REPORT
    assert _rel(sim['u'], u_plain) < 1e-10
    fea.initial_solve = False                    # the hook is bypassed (second branch of fea_dolfinx.py:185)
    sim.run()
    assert len(calls) == 1
    g = np.asarray(sim.compute_totals('l2_functional', 'f'))
    assert np.isfinite(g).all()


@pytest.mark.parametrize("d,n", [(2, 40), (3, 12)])
def test_permuted_numbering(ctx, d, n):
    from femo_amd import engine as E
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    from tests.test_gpu_operators import make_sim
    base = createUnitSquareMesh(n, jitter=0.2) if d == 2 else createUnitCubeMesh(n, jitter=0.2)
    mesh = base.permuted(seed=7, cells=True)
    om = fo.OMesh(d, mesh.x, mesh.conn, n)
    dm = E.DeviceMesh(ctx, mesh.x, mesh.conn)
    assert dm.info["regular_slices"] == 0
    rng = np.random.default_rng(3)
    u, f = rng.standard_normal(mesh.n_vert), rng.standard_normal(mesh.n_cell)
    bd = fo.boundary_vertices_box(mesh.x)
    g = 0.1 * rng.standard_normal(len(bd))
    U, F, R = E.Vec(ctx, mesh.n_vert).set(u), E.Vec(ctx, mesh.n_cell).set(f), E.Vec(ctx, mesh.n_vert)
    E.assemble_residual(dm, 0, None, U, F, R)
    assert _rel(R.get(), fo.residual(om, u, f)) < 1e-12
    K, A, B = E.Mat(dm), E.Mat(dm), E.Vec(ctx, mesh.n_vert)
    ds = E.DirichletSet(dm, bd, g)
    E.assemble_system(dm, 0, None, U, F, ds, K, A, B)
    Kr = fo.stiffness(om)
    Kg = K.to_scipy()
    assert np.array_equal(Kg.indices, Kr.indices) and _rel(Kg.data, Kr.data) < 1e-12
    assert abs(Kg - Kg.T).max() == 0.0                      # bitwise symmetric in any numbering
    assert _rel(A.to_scipy().toarray(), fo.eliminate_bc(Kr, bd).toarray()) < 1e-12
    assert _rel(B.get(), fo.newton_rhs(Kr, fo.residual(om, u, f), u, bd, g)) < 1e-12
    Y = E.Vec(ctx, mesh.n_vert)
    K.mult(U, Y)
    assert _rel(Y.get(), Kr @ u) < 1e-13
    G = E.Vec(ctx, mesh.n_vert)
    ud = fo.u_target(mesh.x)
    E.functional_grad_u(dm, 0, [1e-6], U, F, E.Vec(ctx, mesh.n_vert).set(ud), G)
    assert _rel(G.get(), fo.functional_du(om, u, ud)) < 1e-12
    # whole cycle through the operators; the oracle's result mapped through the permutation equals the base mesh's
    sim, fea, f_ex, _ = make_sim(mesh, device=False)
    fsrc = fo.f_star(fo.centroids(om)) * 0.6 + 0.05
    sim['f'] = fsrc
    sim.run()
    grad = np.asarray(sim.compute_totals('l2_functional', 'f'))
    ref = fo.reference_cycle(om, fsrc, fo.u_target(om.x), bd, np.zeros(len(bd)))
    assert _rel(sim['u'], ref['u']) < 1e-10 and _rel(grad, ref['grad']) < 1e-10


@pytest.mark.parametrize("d,n", [(2, 48), (2, 96), (3, 16), (3, 32)])
def test_bpx_stopping_rule(ctx, d, n):
    """sqrt(r^T M^-1 r) <= rtol sqrt(b^T M^-1 b): the solution error against the direct solve stays within a few
    rtol (relative, max norm) on every size -- the property that lets rtol be chosen without looking at the mesh --
    and Dirichlet rows come out exact however large the lifted values are."""
    import scipy.sparse.linalg as spla
    from femo_amd import engine as E
    m = fo.unit_square_mesh(n, 0.2) if d == 2 else fo.unit_cube_mesh(n, 0.2)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    bd = fo.boundary_vertices_box(m.x)
    ds = E.DirichletSet(dm, bd, 0.0)
    A = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, ds, A)
    rng = np.random.default_rng(n)
    b = fo.load_vector(m, 1.0 + rng.random(m.n_cell))
    b[bd] = 1e3 * rng.standard_normal(len(bd))              # lifted boundary values dwarf the interior right-hand side
    Ar = fo.eliminate_bc(fo.stiffness(m), bd).tocsc()
    x_ref = spla.splu(Ar).solve(b)
    X = E.Vec(ctx, m.n_vert)
    its = {}
    for rtol in (1e-6, 1e-11):
        info = A.solve_cg(E.Vec(ctx, m.n_vert).set(b), X, rtol=rtol, pc="bpx")
        x = np.asarray(X.get())
        interior = np.ones(m.n_vert, bool)
        interior[bd] = False
        assert np.array_equal(x[bd], b[bd])                 # identity rows: exact
        err = np.abs(x - x_ref)[interior].max() / np.abs(x_ref[interior]).max()
        assert err < 3.0 * rtol + 1e-13, (rtol, err)
        assert info.converged == 1 and info.pc_residual_norm <= rtol * info.pc_rhs_norm * 1.0000001
        its[rtol] = info.iterations
    assert its[1e-11] <= 36 and its[1e-6] < its[1e-11]
    # absolute threshold in the same norm: stops before the first iteration when the residual is already below it
    info = A.solve_cg(E.Vec(ctx, m.n_vert).set(b), X, rtol=1e-11, pc="bpx", atol_pc=1e30)
    assert info.iterations == 0 and info.converged == 1


def test_imported_unstructured_mesh_cycle(ctx, tmp_path):
    """A genuinely unstructured mesh entering through import_mesh (utils_dolfinx.py:69-123): L-shaped domain graded
    towards the re-entrant corner, jittered, vertices and cells randomly numbered, Dirichlet data on the two tagged
    boundary parts.  Assembly and the whole operator cycle against the oracle on the same arrays."""
    from femo_amd import engine as E
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction, import_mesh, outputForm, pdeRes, write_mesh_files
    from femo_amd.fea.mesh import Mesh
    from tests.meshes import l_shape_mesh
    utils_hip.set_context(ctx)
    x, conn, edges, tags = l_shape_mesh(24)
    write_mesh_files("lshape", Mesh(x, conn), edges, tags, {"outer": 1, "reentrant": 2}, directory=str(tmp_path))
    mesh, bmf, table = import_mesh(prefix="lshape", dim=2, directory=str(tmp_path))
    om = fo.OMesh(2, mesh.x, mesh.conn, 0)
    dm = mesh.device(ctx)
    assert dm.info["regular_slices"] == 0 and dm.info["max_rowlen"] >= 7
    bd = np.union1d(bmf.vertices(table["outer"]), bmf.vertices(table["reentrant"]))
    # kernels
    rng = np.random.default_rng(2)
    u, f = rng.standard_normal(mesh.n_vert), rng.standard_normal(mesh.n_cell)
    K = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, None, K)
    Kr = fo.stiffness(om)
    Kg = K.to_scipy()
    assert np.array_equal(Kg.indices, Kr.indices) and _rel(Kg.data, Kr.data) < 1e-12 and abs(Kg - Kg.T).max() == 0.0
    R = E.Vec(ctx, mesh.n_vert)
    E.assemble_residual(dm, 0, None, E.Vec(ctx, mesh.n_vert).set(u), E.Vec(ctx, mesh.n_cell).set(f), R)
    assert _rel(R.get(), fo.residual(om, u, f)) < 1e-12
    # operator cycle: f -> u -> J -> dJ/df with Dirichlet values 0 on both tagged parts
    fea = FEA(mesh)
    fea.REPORT = False
    Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
    f_fn, u_fn, ubc, ud = Function(Vf), Function(Vu), Function(Vu), Function(Vu)
    ud_vals = np.sin(np.pi * mesh.x[:, 0]) * np.sin(np.pi * mesh.x[:, 1]) * 0.05
    ud.vector[:] = ud_vals
    ubc.vector.set(0.0)
    fea.add_strong_bc(ubc, [bmf.vertices(table["outer"]), bmf.vertices(table["reentrant"])], Vu)
    fea.add_input('f', f_fn)
    fea.add_state(name='u', function=u_fn, residual_form=pdeRes(u_fn, TestFunction(Vu), f_fn), arguments=['f'])
    fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, ud, 1e-6), arguments=['f', 'u'])
    model = FEAModel(fea=[fea])
    model.create_input('f', shape=mesh.n_cell, val=1.0)
    sim = Simulator(model, device=False)
    fsrc = 1.0 + 0.5 * np.cos(3 * mesh.centroids()[:, 0])
    sim['f'] = fsrc
    sim.run()
    g = np.asarray(sim.compute_totals('l2_functional', 'f'))
    ref = fo.reference_cycle(om, fsrc, ud_vals, bd, np.zeros(len(bd)))
    assert _rel(sim['u'], ref['u']) < 1e-10 and _rel(g, ref['grad']) < 1e-10
    assert abs(sim['l2_functional'][0] - ref['J'][0]) < 1e-10 * abs(ref['J'][0])


def test_poisson_opt_example_reaches_the_analytic_optimum(ctx):
    """BASELINE config 1 / examples/poisson_opt end to end: the optimisation of run_poisson_opt.py (objective scaled by
    1e5, start at f = 0.086, :168-176) driven by the GPU values and adjoint gradients converges to the analytic
    optimal pair the example prints its errors against (Expression_f / Expression_u, :78-92) -- the one pin the
    reference holds for this problem.  The control error is second order in h (8.2e-4 at n = 32, 2.1e-4 at n = 64)."""
    import scipy.optimize as sopt
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitSquareMesh
    from femo_amd.fea.utils_hip import errorNorm
    from tests.test_gpu_operators import make_sim
    utils_hip.set_context(ctx)
    err = {}
    for n in (32, 64):
        mesh = createUnitSquareMesh(n)
        sim, fea, f_ex, u_ex = make_sim(mesh, device=False)

        def fun(f):
            sim['f'] = f
            sim.run()
            return 1e5 * float(sim['l2_functional'][0]), 1e5 * np.array(sim.compute_totals('l2_functional', 'f'))

        f0 = 0.086 * np.ones(fea.inputs_dict['f']['shape'])
        J0 = fun(f0)[0]
        res = sopt.minimize(fun, f0, jac=True, method="L-BFGS-B", options=dict(maxiter=300, ftol=1e-15, gtol=1e-12))
        sim['f'] = res.x
        sim.run()
        err[n] = (errorNorm(f_ex, fea.inputs_dict['f']['function']), errorNorm(u_ex, fea.states_dict['u']['function']), res.fun, J0)
    assert err[64][2] < 1e-3 * err[64][3]                          # 23.7 -> 0.0125
    assert err[64][0] < 3e-4 and err[64][1] < 2e-5
    assert 3.0 < err[32][0] / err[64][0] < 5.0                     # O(h^2)
