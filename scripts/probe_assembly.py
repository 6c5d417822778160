"""Fused F+J assembly pass (femo_assemble_system): time per variant of requested outputs."""
import json
import os
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh

ctx = Context(0)


def run(tag, mesh, reps=5):
    dm = mesh.device(ctx)
    n = mesh.n_vert
    dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
    bc = E.DirichletSet(dm, dofs, 0.1 * np.ones(len(dofs)))
    rng = np.random.default_rng(0)
    f = Vec(ctx, mesh.n_cell).set(1.0 + rng.random(mesh.n_cell))
    u = Vec(ctx, n).set(rng.standard_normal(n))
    out = {"case": tag, "n_dof": n}
    res = {}
    for mode in ("gather",):
        A, J, b = E.Mat(dm), E.Mat(dm), Vec(ctx, n)
        for variant, args in (("A+rhs", (None, A, b)), ("J+A", (J, A, None)), ("rhs", (None, None, b))):
            E.assemble_system(dm, 0, None, u, f, bc, *args)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                E.assemble_system(dm, 0, None, u, f, bc, *args)
            ctx.sync()
            out[f"{mode} {variant} ms"] = (time.perf_counter() - t0) / reps * 1e3
        E.assemble_system(dm, 0, None, u, f, bc, J, A, b)
        ctx.sync()
        if n < 3_000_000:
            res[mode] = (A.to_scipy().data.copy(), J.to_scipy().data.copy(), b.get().copy())
        else:
            d = Vec(ctx, n)
            res[mode] = (A.diagonal(d).get().copy() if hasattr(A, "diagonal") else None, None, b.get().copy())
    print(json.dumps(out), flush=True)


run("square n=300 jitter", createUnitSquareMesh(300, 0.2))
run("cube n=40 jitter", createUnitCubeMesh(40, jitter=0.2))
run("cube n=100", createUnitCubeMesh(100))
run("cube n=215", createUnitCubeMesh(215))
run("cube n=215 jitter", createUnitCubeMesh(215, jitter=0.2))
