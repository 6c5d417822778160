"""How much of the fused assembly pass is cell-numbering locality?  Same mesh, cells renumbered
type-major (all first tets of every cube, then all second tets, ...) so that the lanes of a wave visit
consecutive cells: times of femo_assemble_system for both numberings."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import Mesh, createUnitCubeMesh

ctx = Context(0)
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 215
base = createUnitCubeMesh(n3)
nc = base.n_cell
per = nc // (n3 ** 3)
order = {"cube-major (generator)": np.arange(nc),
         "type-major": np.arange(nc).reshape(-1, per).T.ravel()}
rng = np.random.default_rng(0)
f0 = 1.0 + rng.random(nc)
for name, perm in order.items():
    mesh = Mesh(base.x, np.ascontiguousarray(base.conn[perm]))
    dm = mesh.device(ctx)
    n = mesh.n_vert
    dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
    bc = E.DirichletSet(dm, dofs, 0.1 * np.ones(len(dofs)))
    f = Vec(ctx, nc).set(f0[perm])
    u = Vec(ctx, n).set(rng.standard_normal(n))
    A, J, b = E.Mat(dm), E.Mat(dm), Vec(ctx, n)
    out = {"cells": name}
    for variant, args in (("A+rhs", (None, A, b)), ("J+A", (J, A, None)), ("rhs", (None, None, b))):
        E.assemble_system(dm, 0, None, u, f, bc, *args)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            E.assemble_system(dm, 0, None, u, f, bc, *args)
        ctx.sync()
        out[variant + " ms"] = (time.perf_counter() - t0) / 5 * 1e3
    g = Vec(ctx, n)
    ud = Vec(ctx, n).fill(0.1)
    E.functional_grad_u(dm, 0, [1e-6], u, f, ud, g)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        E.functional_grad_u(dm, 0, [1e-6], u, f, ud, g)
    ctx.sync()
    out["dJ/du ms"] = (time.perf_counter() - t0) / 5 * 1e3
    out["checksum rhs"] = float(np.abs(b.get()).sum())
    print(json.dumps(out), flush=True)
    del A, J, dm, mesh
