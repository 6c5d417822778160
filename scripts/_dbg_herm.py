import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import shell_oracle as so
from femo_amd.engine import Context
from femo_amd.fea.shell import ShellProblem
from tests.test_gpu_shell import E_ROOF, FZ, H_ROOF, NU_ROOF, roof_fixed
ctx = Context(0)
pts, conn = so.scordelis_lo_mesh(32, 32)
V0 = so.ShellSpace(pts, conn)
fixed = roof_fixed(V0)
for trial in range(3):
    for mode in ("lazy", "explicit"):
        prob = ShellProblem(pts, conn, E_ROOF, NU_ROOF, fixed_dofs=fixed, ctx=ctx, pc="lattice")
        if mode == "explicit":
            prob.dev.enable_lattice_pc()
        prob.set_thickness(H_ROOF)
        prob.set_load([0.0, 0.0, FZ])
        w = prob.solve(rtol=1e-10)
        it1 = prob.last_info.iterations
        w = prob.solve(rtol=1e-10)
        print(trial, mode, prob.dev.hermite, it1, prob.last_info.iterations, flush=True)
