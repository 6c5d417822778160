"""Fused F+J pass of the nonlinear Poisson form (+ Nitsche facets) and the mass matrix: ms per launch.
FEMO_LIB=<path> loads another build of the library (A/B of kernel variants)."""
import json
import os
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import _lib
if os.environ.get("FEMO_LIB"):
    _lib.LIB_PATH = os.environ["FEMO_LIB"]
from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
from oracle import femo_oracle as fo

ctx = Context(0)
for tag, mesh in (("square n=1500", createUnitSquareMesh(1500)), ("cube n=120", createUnitCubeMesh(120))):
    dm = mesh.device(ctx)
    om = fo.OMesh(mesh.tdim, mesh.x, mesh.conn)
    dm.set_boundary_facets(fo.boundary_facets(om))
    n = mesh.n_vert
    rng = np.random.default_rng(0)
    u, f, uex = Vec(ctx, n).set(0.3 * rng.standard_normal(n)), Vec(ctx, mesh.n_cell).set(rng.random(mesh.n_cell)), Vec(ctx, n).fill(0.2)
    J, b, M = E.Mat(dm), Vec(ctx, n), E.Mat(dm)
    out = {"case": tag, "lib": os.path.basename(_lib.LIB_PATH)}
    for name, fn in (("NL J+rhs", lambda: E.assemble_system(dm, 1, [10.0], u, f, None, J, None, b, aux=uex)),
                     ("NL J", lambda: E.assemble_jacobian(dm, 1, [10.0], u, f, None, J, aux=uex)),
                     ("mass", lambda: E.assemble_jacobian(dm, 2, None, None, None, None, M))):
        fn(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        ctx.sync()
        out[name + " ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    out["checksum"] = float(np.abs(b.get()).sum())
    print(json.dumps(out), flush=True)
