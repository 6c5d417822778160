"""A few dispatches of the assembly walks at C4 for hardware-counter passes (rocprofv3 --pmc ...):
dR/du + A in one pass, A + Newton rhs, the residual and dJ/du.  PMC serialises dispatches: keep it small."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 215
permute = len(sys.argv) > 2 and sys.argv[2] == "permute"
ctx = Context(0)
mesh = createUnitCubeMesh(n3)
if permute:
    mesh = mesh.permuted(1)
dm = mesh.device(ctx)
n = mesh.n_vert
dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
bc = E.DirichletSet(dm, dofs, np.zeros(len(dofs)))
A, J, b, r, g = E.Mat(dm), E.Mat(dm), Vec(ctx, n), Vec(ctx, n), Vec(ctx, n)
rng = np.random.default_rng(0)
f = Vec(ctx, mesh.n_cell).set(1.0 + rng.random(mesh.n_cell))
u = Vec(ctx, n).set(rng.standard_normal(n))
ud = Vec(ctx, n).set(rng.standard_normal(n))
for _ in range(2):
    E.assemble_system(dm, 0, None, u, f, bc, J, A, None)
    E.assemble_system(dm, 0, None, u, f, bc, None, A, b)
    E.assemble_residual(dm, 0, None, u, f, r)
    E.functional_grad_u(dm, 0, [1e-6], u, f, ud, g)
ctx.sync()
print("done")
