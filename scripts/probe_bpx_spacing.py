"""BPX: effect of the finest lattice spacing (FEMO_BPX_SPACING) on iterations and solve time."""
import json
import os
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh

ctx = Context(0)
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 215
for jitter in (0.0, 0.2):
    for spacing in ("2.0", "1.7", "1.4"):
        os.environ["FEMO_BPX_SPACING"] = spacing
        mesh = createUnitCubeMesh(n3, jitter=jitter)
        dm = mesh.device(ctx)
        n = mesh.n_vert
        dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
        bc = E.DirichletSet(dm, dofs, np.zeros(len(dofs)))
        A = E.Mat(dm)
        f = Vec(ctx, mesh.n_cell).set(1.0 + np.random.default_rng(0).random(mesh.n_cell))
        b = Vec(ctx, n)
        E.assemble_system(dm, 0, None, Vec(ctx, n).fill(0.0), f, bc, None, A, b)
        x = Vec(ctx, n)
        best = None
        for _ in range(3):
            info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
            if best is None or info.solve_ms < best.solve_ms:
                best = info
        print(json.dumps({"n": n3, "jitter": jitter, "spacing": spacing, "its": best.iterations, "ms": best.solve_ms,
                          "ms_per_it": best.solve_ms / best.iterations, "pc": dm.pc_info()}), flush=True)
        del A, dm, mesh
