import numpy as np, scipy.optimize as sopt, sys
sys.path.insert(0, "/root/repo")
from femo_amd.engine import Context
from femo_amd.fea import utils_hip
from femo_amd.fea.mesh import createUnitSquareMesh
from femo_amd.fea.utils_hip import getFuncArray, errorNorm
from tests.test_gpu_operators import make_sim
ctx = Context(0); utils_hip.set_context(ctx)
for n in (32, 64):
    mesh = createUnitSquareMesh(n)
    sim, fea, f_ex, u_ex = make_sim(mesh, device=False)
    nf = fea.inputs_dict['f']['shape']
    def fun(f):
        sim['f'] = f; sim.run()
        J = float(sim['l2_functional'][0]); g = np.array(sim.compute_totals('l2_functional', 'f'))
        return 1e5 * J, 1e5 * g
    f0 = 0.1 * np.ones(nf) * 0.86
    J0 = fun(f0)[0]
    res = sopt.minimize(fun, f0, jac=True, method="L-BFGS-B", options=dict(maxiter=300, ftol=1e-15, gtol=1e-12))
    f_fn = fea.inputs_dict['f']['function']; u_fn = fea.states_dict['u']['function']
    sim['f'] = res.x; sim.run()
    print(n, 'J0', J0, 'J', res.fun, 'its', res.nit, 'control err', errorNorm(f_ex, f_fn), 'state err', errorNorm(u_ex, u_fn),
          'norm f_ex', np.sqrt((getFuncArray(f_ex)**2).mean()))
