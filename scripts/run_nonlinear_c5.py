"""BASELINE config 5 on one GPU: nonlinear Poisson (u^3) with symmetric Nitsche BCs on the
n x n unit square (n = 2236 -> 5,004,169 DOFs), SNES + adjoint gradient through the operator
surface.  Mirrors examples/nonlinear_poisson_opt/run_nonlinear_poisson_opt.py:147-232."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd.csdl_opt.fea_model import FEAModel
from femo_amd.csdl_opt.simulator import Simulator
from femo_amd.engine import Context
from femo_amd.fea import utils_hip
from femo_amd.fea.fea_hip import FEA, Function, FunctionSpace, TestFunction
from femo_amd.fea.mesh import createUnitSquareMesh
from femo_amd.fea.nonlinear_poisson import outputForm, pdeRes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
ctx = Context(0)
utils_hip.set_context(ctx)
t0 = time.perf_counter()
mesh = createUnitSquareMesh(n)
fea = FEA(mesh)
fea.REPORT = False
Vf, Vu = FunctionSpace(mesh, ('DG', 0)), FunctionSpace(mesh, ('CG', 1))
f_fn, u_fn = Function(Vf), Function(Vu)
u_ex = Function(Vu)
u_ex.interpolate(lambda x: np.sin(2 * np.pi * x[0]) * np.sin(np.pi * x[1]))
fea.add_input('f', f_fn)
fea.add_state(name='u', function=u_fn, residual_form=pdeRes(u_fn, TestFunction(Vu), f_fn, u_exact=u_ex, weak_bc=True, sym=True),
              arguments=['f'])
fea.add_output(name='l2_functional', type='scalar', form=outputForm(u_fn, f_fn, u_ex), arguments=['f', 'u'])
fea.PDE_SOLVER = 'SNES'
model = FEAModel(fea=[fea])
model.create_input('f', shape=fea.inputs_dict['f']['shape'], val=0.1)
sim = Simulator(model, device=True)
setup = time.perf_counter() - t0


def cycle():
    u_fn.vector.set(1.0)                      # CSDL's default state value: cold start every cycle
    sim.values['u'].vec.fill(1.0)
    sim.run()
    return sim.compute_totals('l2_functional', 'f')


cycle()
ctx.sync()
del utils_hip.LAST_KSP_INFO[:]
t0 = time.perf_counter()
g = cycle()
ctx.sync()
T = time.perf_counter() - t0
its = [i["iterations"] for i in utils_hip.LAST_KSP_INFO]
print(json.dumps({"workload": f"nonlinear Poisson + Nitsche, unit square n={n}", "n_dof": mesh.n_vert, "n_cell": mesh.n_cell,
                  "cycle_ms": T * 1e3, "dofs_per_s": mesh.n_vert / T, "cg_iterations": its,
                  "newton_linear_solves": len(its) - 1, "J": float(sim['l2_functional'][0]),
                  "cg_ms": sum(i["solve_ms"] for i in utils_hip.LAST_KSP_INFO), "setup_s": setup}))
