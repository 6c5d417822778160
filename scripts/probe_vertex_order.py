"""What would a brick-major vertex numbering buy?  Same cube mesh with the vertices renumbered in
bricks of B^3 (lexicographic inside a brick): times of the fused assembly pass, the SpMV, and a
BPX-CG solve for both numberings."""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
base = createUnitCubeMesh(n3)
np1 = n3 + 1
idx = np.arange(base.n_vert)
i, j, k = idx % np1, (idx // np1) % np1, idx // (np1 * np1)
nb = (np1 + B - 1) // B
brick = ((k // B) * nb + (j // B)) * nb + (i // B)
inner = ((k % B) * B + (j % B)) * B + (i % B)
order = np.lexsort((inner, brick))            # new position -> old vertex
new_of_old = np.empty_like(order)
new_of_old[order] = np.arange(len(order))
meshes = {"lexicographic (generator)": base,
          f"brick-major B={B}": Mesh(base.x[order], new_of_old[base.conn].astype(np.int32))}
rng = np.random.default_rng(0)
for name, mesh in meshes.items():
    dm = mesh.device(ctx)
    n = mesh.n_vert
    dofs = np.nonzero(np.any(np.isclose(mesh.x, 0.0) | np.isclose(mesh.x, 1.0), axis=1))[0]
    bc = E.DirichletSet(dm, dofs, np.zeros(len(dofs)))
    f = Vec(ctx, mesh.n_cell).set(1.0 + rng.random(mesh.n_cell))
    u = Vec(ctx, n).fill(0.0)
    A, J, b = E.Mat(dm), E.Mat(dm), Vec(ctx, n)
    out = {"vertices": name, "regular_slices": dm.info["regular_slices"], "n_slices": dm.info["n_slices"]}
    for variant, args in (("A+rhs", (None, A, b)), ("J+A", (J, A, None))):
        E.assemble_system(dm, 0, None, u, f, bc, *args)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            E.assemble_system(dm, 0, None, u, f, bc, *args)
        ctx.sync()
        out[variant + " ms"] = (time.perf_counter() - t0) / 5 * 1e3
    E.assemble_system(dm, 0, None, u, f, bc, None, A, b)
    xv, yv = Vec(ctx, n).set(rng.standard_normal(n)), Vec(ctx, n)
    out["spmv us"] = min(A.bench_spmv(xv, yv, 50) for _ in range(3)) * 1e3
    x = Vec(ctx, n)
    best = None
    for _ in range(3):
        info = A.solve_cg(b, x, rtol=1e-14, pc="bpx")
        if best is None or info.solve_ms < best.solve_ms:
            best = info
    out.update({"bpx its": best.iterations, "bpx ms": best.solve_ms, "bpx ms/it": best.solve_ms / best.iterations,
                "spmv in loop us": best.spmv_ms / max(best.spmv_samples, 1) * 1e3})
    print(json.dumps(out), flush=True)
    del A, J, dm
