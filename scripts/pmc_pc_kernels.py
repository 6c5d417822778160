"""A handful of dispatches of the PCG kernels at C4 for hardware-counter passes (rocprofv3 --pmc ...):
one BPX solve capped at 3 iterations + one fused assembly pass.  Keep it small: PMC serialises dispatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 215
ctx = Context(0)
if len(sys.argv) > 2 and sys.argv[2] == "square":
    mesh = createUnitSquareMesh(n3)               # the pattern (and SpMV traffic) of BASELINE config 5's n x n square
else:
    mesh = createUnitCubeMesh(n3)
if len(sys.argv) > 2 and sys.argv[2] == "permute":
    mesh = mesh.permuted(seed=20240807)          # bench.py --permute
dm = mesh.device(ctx)
n = mesh.n_vert
dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
bc = E.DirichletSet(dm, dofs, np.zeros(len(dofs)))
A, b = E.Mat(dm), Vec(ctx, n)
f = Vec(ctx, mesh.n_cell).set(1.0 + np.random.default_rng(0).random(mesh.n_cell))
u0 = Vec(ctx, n).fill(0.0)
J, r, g = E.Mat(dm), Vec(ctx, n), Vec(ctx, n)
E.assemble_system(dm, 0, None, u0, f, bc, None, A, b)          # A + Newton rhs (also builds the load vector of f)
E.assemble_system(dm, 0, None, u0, f, bc, J, A, None)          # dR/du + A
E.assemble_residual(dm, 0, None, u0, f, r)
E.functional_grad_u(dm, 0, [1e-6], u0, f, Vec(ctx, n).fill(1.0), g)
x = Vec(ctx, n)
MAX_IT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
info = A.solve_cg(b, x, rtol=1e-14, max_it=MAX_IT, pc="bpx", check_every=MAX_IT)
ctx.sync()
print("iterations", info.iterations)
