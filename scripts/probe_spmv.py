"""SpMV tuning probe: FEMO_SPMV_VARIANT / FEMO_SPMV_BPC are read once per process."""
import os
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd import engine as E
from femo_amd.fea.mesh import createUnitCubeMesh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
ctx = E.Context(0)
mesh = createUnitCubeMesh(n)
dm = E.DeviceMesh(ctx, mesh.x, mesh.conn)
A = E.Mat(dm)
E.assemble_jacobian(dm, 0, None, None, None, None, A)
x = E.Vec(ctx, mesh.n_vert).set(np.random.default_rng(0).standard_normal(mesh.n_vert))
y = E.Vec(ctx, mesh.n_vert)
ms = min(A.bench_spmv(x, y, 50) for _ in range(3))
nnz, N = dm.info["nnz"], mesh.n_vert
B = nnz * 12 + (N + 1) * 4 + 2 * N * 8
print(f"variant={os.environ.get('FEMO_SPMV_VARIANT', '0')} bpc={os.environ.get('FEMO_SPMV_BPC', '8')} n={n}: "
      f"spmv {ms * 1e3:.1f} us -> {B / ms / 1e6:.0f} GB/s algorithmic", flush=True)
