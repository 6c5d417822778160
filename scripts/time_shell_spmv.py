"""Time the shell operator product on the Scordelis-Lo roof.
usage: time_shell_spmv.py [n] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from femo_amd.engine import Context, Vec
from femo_amd.fea.shell import DeviceShell, ShellSpace

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def roof_mesh(nx, nphi, R=25.0, L=25.0, phi_max=np.deg2rad(40.0)):
    xs, ph = np.linspace(0.0, L, nx + 1), np.linspace(0.0, phi_max, nphi + 1)
    X, P = np.meshgrid(xs, ph, indexing="ij")
    pts = np.stack([X.ravel(), R * np.sin(P).ravel(), R * np.cos(P).ravel()], axis=1)
    idx = np.arange((nx + 1) * (nphi + 1)).reshape(nx + 1, nphi + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pts, conn = roof_mesh(n, n)
ctx = Context(0)
dev = DeviceShell(ctx, ShellSpace(pts, conn))
h = Vec(ctx, dev.space.n_vert); h.set(np.full(dev.space.n_vert, 0.25))
vals = Vec(ctx, dev.nnz)
dev.assemble(4.32e8, 0.0, h, vals)
x = Vec(ctx, dev.n_dof); x.set(np.random.default_rng(0).standard_normal(dev.n_dof))
y = Vec(ctx, dev.n_dof)
for _ in range(10):
    dev.matvec(vals, x, y)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    dev.matvec(vals, x, y)
ctx.sync()
us = (time.perf_counter() - t0) / reps * 1e6
nb = dev.nnz / 9
print(f"n_dof {dev.n_dof} nnz {dev.nnz}: "
      f"{us:.1f} us per product, {(8 * dev.nnz + 4 * nb + 16 * dev.n_dof) / us * 1e-6:.2f} TB/s algorithmic")
