"""NumPy/SciPy prototype: vertex-star patch smoothers for the shell preconditioner (CPU).
usage: probe_shell_patch.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from femo_amd.fea.shell import ShellSpace, lattice_pc
from oracle import shell_oracle as so

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cmax = 3200
L_ = 25.0
pts, conn = so.scordelis_lo_mesh(n, n)
V = so.ShellSpace(pts, conn)
S = ShellSpace(pts, conn)
K = so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, 0.0)).tocsr()
F = so.load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1)))
ux, vx = V.unode_x, V.x
on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
fixed = np.unique(np.concatenate([
    V.u_dof(on(ux[:, 0], L_), 1), V.u_dof(on(ux[:, 0], L_), 2), V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0),
    V.theta_dof(on(vx[:, 1], 0.0), 2), V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))
nd = V.n_dof
mask = np.ones(nd); mask[fixed] = 0.0
Dm = sp.diags(mask)
Kf = (Dm @ K @ Dm + sp.diags(1.0 - mask)).tocsr()
b = F * mask
Lp = lattice_pc(S, None)
levels, off = Lp["levels"], Lp["level_offsets"]
nl = len(levels)
rows = np.repeat(np.arange(nd), 8)
P = []
for l in range(nl):
    sl = slice(8 * l, 8 * l + 8)
    Pl = sp.csr_matrix((Lp["ell_w"][:, sl].ravel(), (rows, Lp["ell_idx"][:, sl].ravel() - 6 * off[l])), shape=(nd, 6 * (off[l + 1] - off[l])))
    P.append((Dm @ Pl).tocsr())


def block_diag_inv(A, bs):
    nb = A.shape[0] // bs
    blocks = np.zeros((nb, bs, bs))
    coo = A.tocoo()
    sel = (coo.row // bs) == (coo.col // bs)
    blocks[coo.row[sel] // bs, coo.row[sel] % bs, coo.col[sel] % bs] = coo.data[sel]
    for i in range(nb):
        if not np.any(blocks[i]):
            blocks[i] = np.eye(bs)
    inv = np.linalg.inv(blocks)
    r = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, :, None]) + np.zeros((1, 1, bs), int)
    c = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, None, :]) + np.zeros((1, bs, 1), int)
    return sp.csr_matrix((inv.ravel(), (r.ravel(), c.ravel())), shape=A.shape)


Spt = block_diag_inv(Kf, 3)
A, Binv = [None] * nl, [None] * nl
for l in range(nl):
    Al = (P[l].T @ Kf @ P[l]).tocsr()
    d = Al.diagonal()
    A[l] = (Al + sp.diags((d == 0.0).astype(float))).tocsr()
    Binv[l] = block_diag_inv(A[l], 6)
c = -1
for l in range(nl - 1):
    if A[l].shape[0] <= cmax:
        c = l
lu_c = spla.splu(A[c].tocsc())
print(f"n={n} dofs={nd} levels {levels} coarse level {levels[c]}", flush=True)


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


def lattice_additive(r):
    z = P[c] @ lu_c.solve(P[c].T @ r)
    for l in range(c + 1, nl):
        z += P[l] @ (Binv[l] @ (P[l].T @ r))
    return z


def patch_operator(patches):
    """sum_p R_p^T (R_p K R_p^T)^-1 R_p as a sparse matrix (patches: list of dof index arrays)."""
    rr, cc, vv = [], [], []
    Kc = Kf.tocsr()
    for d in patches:
        B = Kc[d][:, d].toarray()
        Bi = np.linalg.inv(B)
        rr.append(np.repeat(d, d.size)); cc.append(np.tile(d, d.size)); vv.append(Bi.ravel())
    return sp.csr_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))), shape=(nd, nd))


nv, nu = V.n_vert, V.n_unode
ev = S.edge_vertices
edges_of = [[] for _ in range(nv)]
for e, (a, bb) in enumerate(ev):
    edges_of[a].append(e); edges_of[bb].append(e)
three = np.arange(3)


def dofs_u(node):
    return (3 * np.asarray(node)[:, None] + three).ravel()


def dofs_t(vert):
    return (3 * nu + 3 * np.asarray(vert)[:, None] + three).ravel()


star = [np.concatenate([dofs_u([v]), dofs_t([v]), dofs_u(nv + np.asarray(edges_of[v]))]) for v in range(nv)]
node6 = [np.concatenate([dofs_u([v]), dofs_t([v])]) for v in range(nv)] + [dofs_u([nv + e]) for e in range(len(ev))]
# closed star: the vertex, its edges AND the neighbouring vertices' dofs
nbrs = [sorted({int(x) for e in edges_of[v] for x in ev[e]} - {v}) for v in range(nv)]
star2 = [np.concatenate([star[v], dofs_u(nbrs[v]), dofs_t(nbrs[v])]) for v in range(nv)]
# cell patches: all 27 dofs of a cell
cellp = [np.unique(S.cell_dofs[k]) for k in range(S.n_cell)]

variants = {"3x3 point blocks (what runs)": Spt}
t0 = time.time()
variants["6x6 vertex blocks + 3x3 edge blocks"] = patch_operator(node6)
variants["vertex stars (vertex + its edges, ~24 dofs, overlap 2 on edges)"] = patch_operator(star)
variants["cell patches (27 dofs)"] = patch_operator(cellp)
if n <= 32:
    variants["closed vertex stars (~60 dofs)"] = patch_operator(star2)
print(f"patch operators in {time.time()-t0:.1f}s", flush=True)


def lam_max(M, iters=40):
    v = np.random.default_rng(0).standard_normal(nd) * mask
    lam = 1.0
    for _ in range(iters):
        w = M @ (Kf @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    return lam


for name, M in variants.items():
    lm = lam_max(M)
    t0 = time.time()
    it_add = pcg(lambda r: M @ r + lattice_additive(r))
    it_scaled = pcg(lambda r: (M @ r) / lm * 1.0 + lattice_additive(r))
    print(f"{name}: lambda_max(M K) = {lm:.2f}; additive with the lattice: {it_add} its, smoother scaled by 1/lambda_max: {it_scaled} its "
          f"(nnz of M {M.nnz / 1e6:.2f} M vs K {Kf.nnz / 1e6:.2f} M; {time.time()-t0:.1f}s)", flush=True)
