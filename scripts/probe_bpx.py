"""BPX vs Jacobi CG on the device: iteration counts, times, agreement of the solutions."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.engine import Context, Vec
from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh, locate_dofs_geometrical

ctx = Context(0)


def run(tag, mesh, reps=2):
    dm = mesh.device(ctx)
    n = mesh.n_vert
    dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
    bc = E.DirichletSet(dm, dofs, np.zeros(len(dofs)))
    A = E.Mat(dm)
    rng = np.random.default_rng(0)
    f = Vec(ctx, mesh.n_cell).set(1.0 + rng.random(mesh.n_cell))
    u = Vec(ctx, n).fill(0.0)
    b = Vec(ctx, n)
    E.assemble_system(dm, 0, None, u, f, bc, None, A, b)
    out = {"case": tag, "n_dof": n}
    xs = {}
    for pc in ("jacobi", "bpx"):
        x = Vec(ctx, n)
        best = None
        for _ in range(reps):
            info = A.solve_cg(b, x, rtol=1e-14, pc=pc)
            if best is None or info.solve_ms < best.solve_ms:
                best = info
        xs[pc] = x.get()
        out[pc] = dict(its=best.iterations, conv=best.converged, ms=best.solve_ms, res=best.residual_norm,
                       ms_per_it=best.solve_ms / max(best.iterations, 1), spmv_ms=best.spmv_ms / max(best.spmv_samples, 1))
    out["rel_diff"] = float(np.abs(xs["bpx"] - xs["jacobi"]).max() / np.abs(xs["jacobi"]).max())
    out["pc"] = dm.pc_info()
    print(json.dumps(out), flush=True)


for n3 in [int(a) for a in sys.argv[1:]] or [32, 64]:
    run(f"cube n={n3}", createUnitCubeMesh(n3))
    run(f"cube n={n3} jitter", createUnitCubeMesh(n3, jitter=0.2))
run("square n=512", createUnitSquareMesh(512))
