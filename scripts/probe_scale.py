"""Ad-hoc scale probe (not the benchmark): times setup, assembly, SpMV and CG."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import femo_oracle as fo
from femo_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time(); m = fo.unit_cube_mesh(n); t1 = time.time()
print(f"mesh n={n}: {m.n_vert} verts {m.n_cell} cells gen {t1-t0:.2f}s", flush=True)
ctx = E.Context(0)
t0 = time.time(); dm = E.DeviceMesh(ctx, m.x, m.conn); t1 = time.time()
print("device mesh + topology", f"{t1-t0:.2f}s", dm.info, flush=True)
bd = fo.boundary_vertices_box(m.x)
bc = E.DirichletSet(dm, bd, 0.0)
f = fo.f_star(fo.centroids(m))
U, F = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_cell).set(f)
R = E.Vec(ctx, m.n_vert); A = E.Mat(dm); K = E.Mat(dm)
for name, fn in [("residual", lambda: E.assemble_residual(dm, 0, None, U, F, R)),
                 ("jacobian", lambda: E.assemble_jacobian(dm, 0, None, U, F, None, K)),
                 ("jacobian_bc", lambda: E.assemble_jacobian(dm, 0, None, U, F, bc, A))]:
    fn(); ctx.sync(); t0 = time.time()
    for _ in range(5): fn()
    ctx.sync(); print(name, f"{(time.time()-t0)/5*1e3:.3f} ms", flush=True)
B = E.Vec(ctx, m.n_vert); E.newton_rhs(K, R, U, bc, B)
Y = E.Vec(ctx, m.n_vert)
ms = A.bench_spmv(B, Y, 50)
nnz = dm.info["nnz"]; N = m.n_vert
bytes_alg = nnz * 12 + (N + 1) * 4 + 2 * N * 8
print(f"spmv {ms*1e3:.1f} us  alg {bytes_alg/1e6:.1f} MB -> {bytes_alg/ms/1e6:.1f} GB/s", flush=True)
X = E.Vec(ctx, m.n_vert)
for rep in range(2):
    t0 = time.time(); info = A.solve_cg(B, X, rtol=1e-12); t1 = time.time()
    print(f"cg it={info.iterations} conv={info.converged} res={info.residual_norm:.3e} solve_ms={info.solve_ms:.2f} wall={1e3*(t1-t0):.2f} "
          f"per-it={info.solve_ms/max(info.iterations,1)*1e3:.1f} us spmv_sample={info.spmv_ms/max(info.spmv_samples,1)*1e3:.1f} us", flush=True)
