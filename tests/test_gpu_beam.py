"""GPU parity of the Euler-Bernoulli / cubic-Hermite beam (BASELINE config 3 as it exists in the
tree: examples/beam_thickness_opt, 50 elements, 102 DOFs) against the oracle, and against the
golden vector the reference itself holds: the 50 OpenMDAO-optimal thicknesses
(run_thickness_opt_cantilever_beam.py:252-261)."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def build_beam(nel=50, L=1.0, E=1.0, b=0.1, h=0.1, device=False):
    """run_thickness_opt_cantilever_beam.py:41-177 on the HIP mirror."""
    from femo_amd.csdl_opt.fea_model import FEAModel
    from femo_amd.csdl_opt.simulator import Simulator
    from femo_amd.fea.beam import EndpointMeasure, compliance, locate_dofs_at_point, pdeRes, volume
    from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction, createIntervalMesh
    mesh = createIntervalMesh(nel, 0., L)
    fea = FEA(mesh)
    fea.REPORT = False
    Vt = FunctionSpace(mesh, ('DG', 0))
    Vu = FunctionSpace(mesh, ('Hermite', 3))
    t_fn, u_fn = Function(Vt), Function(Vu)
    v = TestFunction(Vu)
    ds_end = EndpointMeasure(mesh)
    f = -1.0
    fea.add_input('thickness', t_fn)
    fea.add_state(name='displacements', function=u_fn, residual_form=pdeRes(u_fn, v, t_fn, f, ds_end, E, b),
                  arguments=['thickness'])
    fea.add_output(name='compliance', type='scalar', form=compliance(u_fn, f, ds_end),
                   arguments=['thickness', 'displacements'])
    fea.add_output(name='volume', type='scalar', form=volume(t_fn, b, L), arguments=['thickness'])
    ubc = Function(Vu)
    ubc.vector.set(0.0)
    fea.add_strong_bc(ubc, locate_dofs_at_point(Vu, 0.0))
    model = FEAModel(fea=[fea])
    model.create_input('thickness', shape=nel, val=h)
    model.add_design_variable('thickness', upper=10., lower=1e-2)
    model.add_objective('compliance')
    return Simulator(model, device=device), fea


def test_beam_kernels_and_cycle(ctx):
    from femo_amd.fea import utils_hip
    utils_hip.set_context(ctx)
    nel, L, E, b, h = 50, 1.0, 1.0, 0.1, 0.1
    rng = np.random.default_rng(4)
    t = h * (1.0 + 0.3 * rng.uniform(-1, 1, nel))
    for device in (False, True):
        sim, fea = build_beam(nel, L, E, b, h, device)
        op = dict(sim.ops)['displacements_state_model']
        # assembled operators vs the oracle
        u_rand = rng.standard_normal(2 * nel + 2)
        res = {}
        op.evaluate_residuals({'thickness': t}, {'displacements': u_rand}, res)
        ref_r = fo.beam_residual(nel, L, u_rand, t, E, b)
        assert np.abs(np.asarray(res['displacements']) - ref_r).max() < 1e-12 * np.abs(ref_r).max()
        op.compute_derivatives({'thickness': t}, {'displacements': u_rand}, {})
        K = fo.beam_stiffness(nel, L, t, E, b)
        assert np.abs(op.dRdu.to_scipy() - K).max() < 1e-12 * abs(K).max()
        D = fo.beam_dRdt(nel, L, u_rand, t, E, b)
        assert np.abs(op.dRdf_dict['thickness']['dRdf'].to_scipy() - D).max() < 1e-12 * abs(D).max()
        # the cycle: cond(K) ~ 1e8, so states agree to eps * cond with either solver
        sim['thickness'] = t
        sim.run()
        ref = fo.beam_cycle(nel, L, t, E, b)
        assert np.abs(sim['displacements'] - ref['u']).max() < 1e-6 * np.abs(ref['u']).max()
        assert abs(sim['compliance'][0] - ref['compliance']) < 1e-6 * ref['compliance']
        assert abs(sim['volume'][0] - ref['volume']) < 1e-14
        g = np.asarray(sim.compute_totals('compliance', 'thickness'))
        assert np.abs(g - ref['grad_compliance']).max() < 1e-5 * np.abs(ref['grad_compliance']).max()
        gv = np.asarray(sim.compute_totals('volume', 'thickness'))
        assert np.abs(gv - ref['grad_volume']).max() < 1e-15
    # closed form: uniform thickness, tip deflection P L^3 / (3 E I) (Hermite elements are nodally exact)
    sim['thickness'] = np.full(nel, h)
    sim.run()
    tip = sim['displacements'][2 * nel]
    assert abs(tip + L ** 3 / (3 * E * b * h ** 3 / 12)) < 1e-6 * abs(tip)


def test_beam_optimum_matches_reference_golden_vector(ctx):
    """SLSQP on (compliance, volume) with the GPU adjoint gradients reproduces the thickness
    distribution the reference prints as its cross-check (run_thickness_opt_cantilever_beam.py:
    191, 252-261: SLSQP, ftol 1e-9; 50 values)."""
    import scipy.optimize as so
    from femo_amd.fea import utils_hip
    utils_hip.set_context(ctx)
    nel, L, b, h = 50, 1.0, 0.1, 0.1
    sim, fea = build_beam(nel, L, 1.0, b, h, device=False)
    cache = {}

    def evaluate(t):
        key = t.tobytes()
        if key not in cache:
            cache.clear()
            sim['thickness'] = t
            sim.run()
            cache[key] = (float(sim['compliance'][0]), np.asarray(sim.compute_totals('compliance', 'thickness')).copy(),
                          float(sim['volume'][0]), np.asarray(sim.compute_totals('volume', 'thickness')).copy())
        return cache[key]

    res = so.minimize(lambda t: evaluate(t)[0], np.full(nel, h), jac=lambda t: evaluate(t)[1], method='SLSQP',
                      bounds=[(1e-2, 10.)] * nel, options={'maxiter': 1000, 'ftol': 1e-12},
                      constraints=[{'type': 'eq', 'fun': lambda t: evaluate(t)[2] - b * h * L, 'jac': lambda t: evaluate(t)[3]}])
    assert res.success
    assert np.abs(res.x - fo.BEAM_THICK_REF).max() < 1e-5
    assert abs(res.fun - 23762.153677) < 1e-2
