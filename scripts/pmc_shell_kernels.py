"""A handful of dispatches of the shell kernels on the n x n roof for hardware-counter passes (rocprofv3 --pmc ...):
assembly, penalty-free solve capped at a few iterations (FEMO_SHELL_PMC_ITS), compliance, dK/dh contraction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from femo_amd import _lib
from femo_amd.engine import Context
from femo_amd.fea.shell import ShellProblem, ShellSpace
from oracle import shell_oracle as so

n = int(sys.argv[1]) if len(sys.argv) > 1 else 362
its = int(os.environ.get("FEMO_SHELL_PMC_ITS", "4"))
L = 25.0
pts, conn = so.scordelis_lo_mesh(n, n, L=L)
S = ShellSpace(pts, conn)
on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
ux, vx = S.unode_x, S.x
fixed = np.unique(np.concatenate([
    S.u_dof(on(ux[:, 0], L), 1), S.u_dof(on(ux[:, 0], L), 2), S.u_dof(on(ux[:, 1], 0.0), 1), S.theta_dof(on(vx[:, 1], 0.0), 0),
    S.theta_dof(on(vx[:, 1], 0.0), 2), S.u_dof(on(ux[:, 0], 0.0), 0), S.theta_dof(on(vx[:, 0], 0.0), 1), S.theta_dof(on(vx[:, 0], 0.0), 2)]))
ctx = Context(0)
prob = ShellProblem(pts, conn, 4.32e8, 0.0, fixed_dofs=fixed, ctx=ctx)
prob.dev.enable_lattice_pc()
prob.set_thickness(0.25)
prob.set_load([0.0, 0.0, -90.0])
K = prob._stiffness()
prob.dev.load(prob.f, prob.F)
try:
    prob.dev.solve(K, prob.F, prob.w, fixed=prob.fixed, rtol=1e-10, max_it=its, check_every=its)
except Exception as e:            # not converged after `its` iterations: expected
    print("solve stopped:", str(e)[:60])
J = prob.dev.compliance(prob.w, grad=prob.tmp)
prob.dev.dform_dh(prob.E, prob.nu, prob.h, prob.tmp, prob.w, out=prob.gh)
ctx.sync()
print("done", S.n_dof)
