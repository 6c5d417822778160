"""NumPy/SciPy prototype: lattice spaces that mirror the element (tri-QUADRATIC displacements, trilinear rotations) against
the trilinear / trilinear spaces that run (CPU).   usage: probe_shell_q2.py [n] [thickness]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from femo_amd.fea.shell import ShellSpace
from oracle import shell_oracle as so

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
thick = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
cmax = 3200
L_ = 25.0
pts, conn = so.scordelis_lo_mesh(n, n)
V = so.ShellSpace(pts, conn)
S = ShellSpace(pts, conn)
K = so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, thick), 4.32e8, 0.0)).tocsr()
F = so.load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1)))
ux, vx = V.unode_x, V.x
on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
fixed = np.unique(np.concatenate([
    V.u_dof(on(ux[:, 0], L_), 1), V.u_dof(on(ux[:, 0], L_), 2), V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0),
    V.theta_dof(on(vx[:, 1], 0.0), 2), V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))
nd, nu, nv = V.n_dof, V.n_unode, V.n_vert
mask = np.ones(nd); mask[fixed] = 0.0
Dm = sp.diags(mask)
Kf = (Dm @ K @ Dm + sp.diags(1.0 - mask)).tocsr()
b = F * mask
allp = np.concatenate([S.unode_x, S.x])
lo = allp.min(axis=0)
ext = float((allp.max(axis=0) - lo).max()) * (1.0 + 1e-9)
h_avg = float(np.linalg.norm(S.x[S.edge_vertices[:, 0]] - S.x[S.edge_vertices[:, 1]], axis=1).mean())
lg = np.log2(max(ext / h_avg, 2.0))
finest = max(2, 2 ** int(np.floor(lg - 1.0 + 1e-9)), min(2 ** int(round(lg)), 32))
levels = [m for m in (2, 4, 8, 16, 32, 64, 128, 256) if m <= finest]
print(f"n={n} dofs={nd} t={thick} levels {levels}", flush=True)


def interp(points, m, order):
    """(node ids on the compacted lattice, weights, n_nodes): order 1 = trilinear on m cells per axis, order 2 = triquadratic
    (nodes on the 2 m + 1 grid, Lagrange basis on each cell's 3 x 3 x 3 nodes)."""
    t = (points - lo) / ext * m
    i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
    fr = t - i0
    if order == 1:
        w1 = [np.stack([1.0 - fr[:, k], fr[:, k]], axis=1) for k in range(3)]
        base, stride, npa = i0, 1, 2
        N1 = m + 1
    else:
        x = fr
        w1 = [np.stack([(1 - x[:, k]) * (1 - 2 * x[:, k]), 4 * x[:, k] * (1 - x[:, k]), x[:, k] * (2 * x[:, k] - 1)], axis=1) for k in range(3)]
        base, stride, npa = 2 * i0, 1, 3
        N1 = 2 * m + 1
    ids, ws = [], []
    for a in range(npa):
        for bq in range(npa):
            for cq in range(npa):
                ids.append(((base[:, 2] + cq) * N1 + base[:, 1] + bq) * N1 + base[:, 0] + a)
                ws.append(w1[0][:, a] * w1[1][:, bq] * w1[2][:, cq])
    ids, ws = np.stack(ids, axis=1), np.stack(ws, axis=1)
    uniq, inv = np.unique(ids.ravel(), return_inverse=True)
    return inv.reshape(ids.shape), ws, uniq.size


def prolongation(m, order_u):
    """P for level m: displacement fields with `order_u`, rotations trilinear; unknowns 3 per displacement node, then 3 per rotation node."""
    iu, wu, nnu = interp(S.unode_x, m, order_u)
    it, wt, nnt = interp(S.x, m, 1)
    rows, cols, vals = [], [], []
    for k in range(3):
        rows.append(np.repeat(3 * np.arange(nu) + k, iu.shape[1])); cols.append((3 * iu + k).ravel()); vals.append(wu.ravel())
        rows.append(np.repeat(3 * nu + 3 * np.arange(nv) + k, it.shape[1])); cols.append((3 * nnu + 3 * it + k).ravel()); vals.append(wt.ravel())
    Pm = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, 3 * nnu + 3 * nnt))
    return (Dm @ Pm).tocsr()


def block_diag_inv(A, bs):
    nb = A.shape[0] // bs
    blocks = np.zeros((nb, bs, bs))
    coo = A.tocoo()
    sel = (coo.row // bs) == (coo.col // bs)
    blocks[coo.row[sel] // bs, coo.row[sel] % bs, coo.col[sel] % bs] = coo.data[sel]
    for i in range(nb):
        d = np.diag(blocks[i])
        if not np.all(d > 0):
            bad = ~(d > 0)
            blocks[i][bad, :] = 0.0; blocks[i][:, bad] = 0.0
            blocks[i][bad, bad] = 1.0
    inv = np.linalg.inv(blocks)
    r = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, :, None]) + np.zeros((1, 1, bs), int)
    c = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, None, :]) + np.zeros((1, bs, 1), int)
    return sp.csr_matrix((inv.ravel(), (r.ravel(), c.ravel())), shape=A.shape)


Spt = block_diag_inv(Kf, 3)


def pcg(apply_pc, rtol=1e-10, maxit=3000):
    x = np.zeros(nd); r = b.copy(); z = apply_pc(r); p = z.copy(); g = r @ z; g0 = g
    for it in range(1, maxit + 1):
        q = Kf @ p
        a = g / (p @ q)
        x += a * p; r -= a * q
        z = apply_pc(r); g1 = r @ z
        if g1 <= rtol ** 2 * g0:
            return it
        p = z + (g1 / g) * p; g = g1
    return maxit


def build(order_u, coarse_cap):
    P = [prolongation(m, order_u) for m in levels]
    A = []
    for Pl in P:
        Al = (Pl.T @ Kf @ Pl).tocsr()
        A.append((Al + sp.diags((Al.diagonal() == 0.0).astype(float))).tocsr())
    c = 0
    for l in range(len(levels) - 1):
        if A[l].shape[0] <= coarse_cap:
            c = l
    lu = spla.splu(A[c].tocsc())
    Binv = [block_diag_inv(A[l], 3) if l > c else None for l in range(len(levels))]

    def apply(r):
        z = Spt @ r + P[c] @ lu.solve(P[c].T @ r)
        for l in range(c + 1, len(levels)):
            z += P[l] @ (Binv[l] @ (P[l].T @ r))
        return z
    return apply, levels[c], [a.shape[0] for a in A]


for order_u, cap in ((1, cmax), (2, cmax), (2, 4 * cmax)):
    t0 = time.time()
    ap, clev, sizes = build(order_u, cap)
    its = pcg(ap)
    print(f"displacements order {order_u} (rotations trilinear), exact solve on level {clev} (cap {cap}), unknowns per level {sizes}: "
          f"{its} iterations ({time.time()-t0:.1f}s)", flush=True)
