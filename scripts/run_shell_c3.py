"""BASELINE config 3 on the HIP shell path: Scordelis-Lo roof at scale -- stiffness assembly, forward solve, compliance,
adjoint solve, thickness sensitivity.  One JSON line (profiles/rNN_config3_shell_roof_nN.json).
usage: run_shell_c3.py [n]   (n x n x 2 triangles)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from femo_amd.engine import Context
from femo_amd.fea.shell import ShellProblem, ShellSpace


def roof_mesh(nx, nphi, R=25.0, L=25.0, phi_max=np.deg2rad(40.0)):
    xs, ph = np.linspace(0.0, L, nx + 1), np.linspace(0.0, phi_max, nphi + 1)
    X, P = np.meshgrid(xs, ph, indexing="ij")
    pts = np.stack([X.ravel(), R * np.sin(P).ravel(), R * np.cos(P).ravel()], axis=1)
    idx = np.arange((nx + 1) * (nphi + 1)).reshape(nx + 1, nphi + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])


n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PC = sys.argv[2] if len(sys.argv) > 2 else "lattice"
L = 25.0
pts, conn = roof_mesh(n, n)
t0 = time.perf_counter()
S = ShellSpace(pts, conn)
on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
ux, vx = S.unode_x, S.x
fixed = np.unique(np.concatenate([
    S.u_dof(on(ux[:, 0], L), 1), S.u_dof(on(ux[:, 0], L), 2), S.u_dof(on(ux[:, 1], 0.0), 1), S.theta_dof(on(vx[:, 1], 0.0), 0),
    S.theta_dof(on(vx[:, 1], 0.0), 2), S.u_dof(on(ux[:, 0], 0.0), 0), S.theta_dof(on(vx[:, 0], 0.0), 1), S.theta_dof(on(vx[:, 0], 0.0), 2)]))
ctx = Context(0)
prob = ShellProblem(pts, conn, 4.32e8, 0.0, fixed_dofs=fixed, ctx=ctx, pc=PC)
setup_s = time.perf_counter() - t0
t0 = time.perf_counter()
if PC == "lattice":
    prob.dev.enable_lattice_pc(int(os.environ["FEMO_SHELL_FINEST"]) if "FEMO_SHELL_FINEST" in os.environ else None)
pc_setup_s = time.perf_counter() - t0
prob.set_thickness(0.25)
prob.set_load([0.0, 0.0, -90.0])
ctx.sync()
t0 = time.perf_counter(); prob._stiffness(); ctx.sync(); t_asm = time.perf_counter() - t0
prob._K_for = None
t0 = time.perf_counter(); prob._stiffness(); ctx.sync(); t_asm = min(t_asm, time.perf_counter() - t0)
t0 = time.perf_counter(); w = prob.solve(rtol=1e-10); t_fwd = time.perf_counter() - t0
it_fwd, ms_fwd = prob.last_info.iterations, prob.last_info.solve_ms
t0 = time.perf_counter()
J, dJdw = prob.compliance(grad=True)
dJdw[prob.fixed.astype(bool)] = 0.0
lam = prob.solve_adjoint(dJdw, rtol=1e-10)
g = -prob.dRdh_T(lam)
t_adj = time.perf_counter() - t0
it_adj, ms_adj = prob.last_info.iterations, prob.last_info.solve_ms
tip = int(np.argmin(np.abs(vx[:, 0]) + np.abs(vx[:, 1] - vx[:, 1].max())))
print(json.dumps({
    "workload": f"Scordelis-Lo roof {n} x {n} x 2 triangles, CG2^3 x CG1^3 Reissner-Mindlin shell: assemble K(h), solve K w = F, "
                "compliance, adjoint solve, dJ/dh (thickness sensitivity)",
    "preconditioner": PC, "pc_levels": prob.dev.pc_levels, "pc_coarse_solve_level": getattr(prob.dev, "coarse_level", None), "pc_setup_s": pc_setup_s,
    "n_dof": int(S.n_dof), "n_cell": int(S.n_cell), "nnz": int(prob.dev.nnz), "setup_s": setup_s,
    "assemble_ms": t_asm * 1e3, "forward_solve_s": t_fwd, "forward_cg_iterations": int(it_fwd), "forward_solve_device_ms": ms_fwd,
    "adjoint_s": t_adj, "adjoint_cg_iterations": int(it_adj), "adjoint_solve_device_ms": ms_adj,
    "ms_per_cg_iteration": ms_fwd / max(it_fwd, 1), "tip_deflection": float(S.vertex_displacement(w)[tip, 2]),
    "tip_reference": -0.3024, "compliance": float(J), "grad_norm": float(np.linalg.norm(g)),
    "dofs_per_s_cycle": S.n_dof / (t_asm + t_fwd + t_adj)}))
