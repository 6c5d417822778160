"""GPU parity of field outputs: L2 projection onto CG1 (utils_dolfinx.py:549-583,
output_model.py:90-159) against the CPU oracle."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - b).max() / np.abs(b).max()


@pytest.mark.parametrize("d,n,jit", [(2, 17, 0.2), (3, 7, 0.2)])
def test_project_matches_oracle(ctx, d, n, jit):
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import Function, FunctionSpace, GradientMagnitude, PowerExpr, project
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    utils_hip.set_context(ctx)
    mesh = createUnitSquareMesh(n, jit) if d == 2 else createUnitCubeMesh(n, jit)
    om = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    rng = np.random.default_rng(8)
    Vu, Vf = FunctionSpace(mesh, ('CG', 1)), FunctionSpace(mesh, ('DG', 0))
    u, w, out = Function(Vu), Function(Vf), Function(Vu)
    un, wc = rng.standard_normal(om.n_vert), rng.uniform(0.5, 1.5, om.n_cell)
    u.vector[:] = un
    w.vector[:] = wc
    project(w, out)                                            # DG0 -> CG1
    assert _rel(out.vector.getArray(), fo.project_l2(om, cell_values=wc)) < 1e-10
    project(w, out, lump_mass=True)
    assert _rel(out.vector.getArray(), fo.project_l2(om, cell_values=wc, lump_mass=True)) < 1e-12
    project(u, out)                                            # CG1 -> CG1 is the identity
    assert _rel(out.vector.getArray(), un) < 1e-10
    project(u, out, lump_mass=True)
    assert _rel(out.vector.getArray(), fo.project_l2(om, nodal_values=un, lump_mass=True)) < 1e-12
    project(GradientMagnitude(u), out)
    assert _rel(out.vector.getArray(), fo.project_l2(om, cell_values=fo.grad_magnitude(om, un))) < 1e-10
    project(PowerExpr(w, 3.0), out)                            # run_topo_opt_cantilever_beam.py:257
    assert _rel(out.vector.getArray(), fo.project_l2(om, cell_values=wc ** 3)) < 1e-10
    # the mass matrix itself
    from femo_amd import engine as E
    M = E.Mat(mesh.device(ctx))
    E.assemble_jacobian(mesh.device(ctx), 2, None, None, None, None, M)
    Mo = fo.mass_matrix(om)
    assert np.array_equal(M.to_scipy().indices, Mo.indices) and _rel(M.to_scipy().data, Mo.data) < 1e-13


def test_field_output_operation(ctx):
    """FEAModel wires an OutputFieldModel per field output ('{name}_output_model', fea_model.py:31-38)."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import GradientMagnitude
    from femo_amd.fea.mesh import createUnitSquareMesh
    from test_gpu_operators import build_poisson
    utils_hip.set_context(ctx)
    n = 16
    mesh = createUnitSquareMesh(n, 0.2)
    om = fo.unit_square_mesh(n, 0.2)
    for device in (False, True):
        fea, f_ex, u_ex = build_poisson(mesh)
        fea.add_field_output('grad_u_mag', GradientMagnitude(fea.states_dict['u']['function']), ['u'])
        model = FEAModel(fea=[fea])
        model.create_input('f', shape=mesh.n_cell, val=0.086)
        sim = Simulator(model, device=device)
        assert [nm for nm, _ in sim.ops] == ['u_state_model', 'l2_functional_output_model', 'grad_u_mag_output_model']
        sim.run()
        bd = fo.boundary_vertices_box(om.x)
        u, _ = fo.newton_solve(om, 0.086 * np.ones(om.n_cell), np.ones(om.n_vert), bd, np.zeros(len(bd)))
        ref = fo.project_l2(om, cell_values=fo.grad_magnitude(om, u))
        assert sim['grad_u_mag'].shape == (om.n_vert,) and _rel(sim['grad_u_mag'], ref) < 1e-9


@pytest.mark.parametrize("d,n", [(2, 9), (3, 5)])
def test_error_norms(ctx, d, n):
    """errorNorm (utils_dolfinx.py:225-238), L2 and H1, against the oracle's mass and stiffness matrices:
    ||e||_L2^2 = e^T M e, |e|_H1^2 = e^T K e for P1 functions."""
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import Function, FunctionSpace, errorNorm, setFuncArray
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    utils_hip.set_context(ctx)
    mesh = createUnitSquareMesh(n, 0.2) if d == 2 else createUnitCubeMesh(n, 0.2)
    om = fo.unit_square_mesh(n, 0.2) if d == 2 else fo.unit_cube_mesh(n, 0.2)
    V = FunctionSpace(mesh, ('CG', 1))
    a, b = Function(V), Function(V)
    rng = np.random.default_rng(4)
    va, vb = rng.standard_normal(mesh.n_vert), rng.standard_normal(mesh.n_vert)
    setFuncArray(a, va); setFuncArray(b, vb)
    e = va - vb
    l2 = np.sqrt(e @ (fo.mass_matrix(om) @ e))
    h1 = np.sqrt(l2 ** 2 + e @ (fo.stiffness(om) @ e))
    assert abs(errorNorm(a, b) - l2) < 1e-12 * l2
    assert abs(errorNorm(a, b, norm='H1') - h1) < 1e-12 * h1
    with pytest.raises(NotImplementedError):
        errorNorm(a, b, norm='Linf')
