import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from oracle import femo_oracle as fo
from femo_amd.fea import utils_hip
from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
from test_gpu_operators import make_sim
for d, n in [(2, 64), (3, 24), (3, 40)]:
    mesh = createUnitSquareMesh(n) if d == 2 else createUnitCubeMesh(n)
    om = fo.unit_square_mesh(n) if d == 2 else fo.unit_cube_mesh(n)
    bd = fo.boundary_vertices_box(om.x)
    f = 0.086 * (1.0 + 0.3 * np.random.default_rng(11).uniform(-1, 1, om.n_cell))
    ref = fo.reference_cycle(om, f, fo.u_target(om.x), bd, np.zeros(len(bd)))
    for rtol in [1e-12, 1e-13, 1e-14, 1e-15, 1e-16]:
        utils_hip.KSP_OPTIONS["rtol"] = rtol
        sim, fea, f_ex, u_ex = make_sim(mesh, True)
        sim['f'] = f
        try:
            sim.run()
            g = np.asarray(sim.compute_totals('l2_functional', 'f'))
        except Exception as e:
            print(d, n, rtol, "FAILED", e); continue
        its = [k["iterations"] for k in utils_hip.LAST_KSP_INFO[-4:]]
        eu = np.abs(sim['u'] - ref['u']).max() / np.abs(ref['u']).max()
        eg = np.abs(g - ref['grad']).max() / np.abs(ref['grad']).max()
        print(f"d={d} n={n} rtol={rtol:g} state_err={eu:.2e} grad_err={eg:.2e} its={its}", flush=True)
