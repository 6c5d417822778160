"""NumPy/SciPy prototype: lattice coarse spaces whose displacement interpolation uses the nodal ROTATIONS as slopes
(Hermite-type, Kirchhoff-conforming for inextensional bending) against the trilinear ones that run.
  u(x)     = sum_n [ H0_n(x) U_n + sum_j H1_{n,j}(x) (Theta_n x e_j) ]     H0 = prod_k h0(t_k),  H1_j = h1(t_j) prod_{k != j} h0(t_k)
  theta(x) = sum_n phi_n(x) Theta_n                                        (trilinear)
with the cubic Hermite pair h0(t) = 1 - 3 t^2 + 2 t^3, h1(t) = t (1 - t)^2 H per axis (H = lattice spacing).
usage: probe_shell_hermite.py [n]"""
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
RIDGE = float(os.environ.get('RIDGE', '1e-8'))
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


def corner_data(points, m):
    t = (points - lo) / ext * m
    i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
    return i0, t - i0, ext / m


def prolongation(m, hermite):
    """Unknowns: 6 per lattice node (U, Theta) on the compacted set of nodes any point touches."""
    i0u, fu, H = corner_data(S.unode_x, m)
    i0t, ft, _ = corner_data(S.x, m)
    ids_all = []
    for i0 in (i0u, i0t):
        for c in range(8):
            bits = [(c >> k) & 1 for k in range(3)]
            ids_all.append(((i0[:, 2] + bits[2]) * (m + 1) + i0[:, 1] + bits[1]) * (m + 1) + i0[:, 0] + bits[0])
    uniq = np.unique(np.concatenate(ids_all))
    rows, cols, vals = [], [], []

    def add(r, node_gid, field, w):
        rows.append(r); cols.append(6 * np.searchsorted(uniq, node_gid) + field); vals.append(w)

    # 1-D shapes at the point for corner bit 0 / 1 of an axis
    def shapes(fr):
        lin = [1.0 - fr, fr]
        if not hermite:
            return lin, lin, None
        h0 = [1.0 - 3 * fr ** 2 + 2 * fr ** 3, 3 * fr ** 2 - 2 * fr ** 3]
        # slope shape: derivative 1 at the node, value 0 at both ends; the offset x - x_n it multiplies is signed
        h1 = [fr * (1.0 - fr) ** 2 * H, -(1.0 - fr) * fr ** 2 * H]
        return lin, h0, h1

    eps = {(0, 1, 2): 1.0, (1, 2, 0): 1.0, (2, 0, 1): 1.0, (0, 2, 1): -1.0, (2, 1, 0): -1.0, (1, 0, 2): -1.0}
    linu, h0u, h1u = zip(*[shapes(fu[:, k]) for k in range(3)])
    lint, _, _ = zip(*[shapes(ft[:, k]) for k in range(3)])
    pu = np.arange(nu)
    pt = np.arange(nv)
    for c in range(8):
        bits = [(c >> k) & 1 for k in range(3)]
        gid_u = ((i0u[:, 2] + bits[2]) * (m + 1) + i0u[:, 1] + bits[1]) * (m + 1) + i0u[:, 0] + bits[0]
        gid_t = ((i0t[:, 2] + bits[2]) * (m + 1) + i0t[:, 1] + bits[1]) * (m + 1) + i0t[:, 0] + bits[0]
        w0 = h0u[0][bits[0]] * h0u[1][bits[1]] * h0u[2][bits[2]]
        wt = lint[0][bits[0]] * lint[1][bits[1]] * lint[2][bits[2]]
        for i in range(3):
            add(3 * pu + i, gid_u, i, w0)                                    # u_i <- U_i
            add(3 * nu + 3 * pt + i, gid_t, 3 + i, wt)                       # theta_i <- Theta_i
        if hermite:
            for j in range(3):                                               # slope along axis j
                w1 = h1u[j][bits[j]]
                for k in range(3):
                    if k != j:
                        w1 = w1 * h0u[k][bits[k]]
                for (i, kk, jj), sgn in eps.items():                         # u_i += (Theta x e_j)_i = eps_{i kk j} Theta_kk
                    if jj == j:
                        add(3 * pu + i, gid_u, 3 + kk, sgn * w1)
    Pm = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, 6 * uniq.size))
    Pm.sum_duplicates()
    return (Dm @ Pm).tocsr()


def block_diag_inv(A, bs):
    nb = A.shape[0] // bs
    blocks = np.zeros((nb, bs, bs))
    coo = A.tocoo()
    sel = (coo.row // bs) == (coo.col // bs)
    blocks[coo.row[sel] // bs, coo.row[sel] % bs, coo.col[sel] % bs] = coo.data[sel]
    for i in range(nb):
        d = np.diag(blocks[i]).copy()
        bad = ~(d > 0)
        if bad.any():
            blocks[i][bad, :] = 0.0; blocks[i][:, bad] = 0.0
            blocks[i][bad, bad] = 1.0
    inv = np.linalg.pinv(blocks, rcond=1e-10, hermitian=True)      # nodes few points touch have rank-deficient blocks
    r = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, :, None]) + np.zeros((1, 1, bs), int)
    c = (np.arange(nb)[:, None, None] * bs + np.arange(bs)[None, None, :]) + np.zeros((1, bs, 1), int)
    return sp.csr_matrix((inv.ravel(), (r.ravel(), c.ravel())), shape=A.shape)


Spt = block_diag_inv(Kf, 3)


x_ref = spla.spsolve(Kf.tocsc(), b)


def pcg(apply_pc, rtol=1e-10, maxit=3000):
    """Returns (iterations, relative error of the iterate against the direct solve, smallest r.z seen): a preconditioner
    that is not positive definite shows up as r.z <= 0 (reported as -iterations)."""
    x = np.zeros(nd); r = b.copy(); z = apply_pc(r); p = z.copy(); g = r @ z; g0 = g
    for it in range(1, maxit + 1):
        q = Kf @ p
        a = g / (p @ q)
        x += a * p; r -= a * q
        z = apply_pc(r); g1 = r @ z
        if not g1 > 0.0:
            return -it, float(np.abs(x - x_ref).max() / np.abs(x_ref).max())
        if g1 <= rtol ** 2 * g0:
            return it, float(np.abs(x - x_ref).max() / np.abs(x_ref).max())
        p = z + (g1 / g) * p; g = g1
    return maxit, float(np.abs(x - x_ref).max() / np.abs(x_ref).max())


def build(hermite, exact_level=None):
    P = [prolongation(m, hermite) for m in levels]
    A = []
    for Pl in P:
        Al = (Pl.T @ Kf @ Pl).tocsr()
        d = Al.diagonal()
        # lattice unknowns no free dof reaches get a unit diagonal; a relative ridge keeps the operator definite where the
        # six unknowns of a node are (nearly) dependent on the surface (RIDGE from the environment, default 1e-8)
        A.append((Al + sp.diags(np.where(d > 0.0, RIDGE * d, 1.0))).tocsr())
    c = 0
    for l in range(len(levels) - 1):
        if A[l].shape[0] <= cmax:
            c = l
    if exact_level is not None:
        c = exact_level
    lu = spla.splu(A[c].tocsc())
    Binv = [block_diag_inv(A[l], 6) if l > c else None for l in range(len(levels))]

    def apply(r):
        z = Spt @ r + P[c] @ lu.solve(P[c].T @ r)
        for l in range(c + 1, len(levels)):
            z += P[l] @ (Binv[l] @ (P[l].T @ r))
        return z
    return apply, levels[c], [a.shape[0] for a in A], P, A


for hermite in (False, True):
    t0 = time.time()
    ap, clev, sizes, P, A = build(hermite)
    its, err = pcg(ap)
    print(f"{'Hermite (rotations as slopes)' if hermite else 'trilinear'}: exact solve on level {clev}, unknowns per level {sizes}: {its} iterations, "
          f"error {err:.1e} ({time.time()-t0:.1f}s)", flush=True)
    # two-grid quality of the finest lattice alone: exact solve there + point blocks, additive
    lu_f = spla.splu(A[-1].tocsc())
    its2, err2 = pcg(lambda r: Spt @ r + P[-1] @ lu_f.solve(P[-1].T @ r))
    print(f"   two-level (exact on the finest lattice, {sizes[-1]} unknowns) + point blocks, additive: {its2} iterations, error {err2:.1e}", flush=True)


# ---- nested variant: only the finest lattice is interpolated from the mesh; coarser levels through lattice-to-lattice
# transfers of the nodal (U, Theta) -- what the GPU hierarchy needs (one mesh-level transfer per iteration) ----------------
def node_sets():
    out = []
    for m in levels:
        ids = []
        for pts_ in (S.unode_x, S.x):
            i0, _, _ = corner_data(pts_, m)
            for c in range(8):
                bits = [(c >> k) & 1 for k in range(3)]
                ids.append(((i0[:, 2] + bits[2]) * (m + 1) + i0[:, 1] + bits[1]) * (m + 1) + i0[:, 0] + bits[0])
        out.append(np.unique(np.concatenate(ids)))
    return out


def lattice_transfer(l, sets, hermite):
    """T: unknowns of level l (coarse) -> unknowns of level l + 1 (fine): the coarse field evaluated at the fine lattice's nodes."""
    mc, mf = levels[l], levels[l + 1]
    g = sets[l + 1]
    ijk = np.stack([g % (mf + 1), (g // (mf + 1)) % (mf + 1), g // ((mf + 1) ** 2)], axis=1)
    Hc = ext / mc
    xf = ijk / mf * ext + lo                                       # positions of the fine nodes
    t = ijk / 2.0                                                  # in coarse cell units
    i0 = np.clip(np.floor(t).astype(np.int64), 0, mc - 1)
    fr = t - i0
    rows, cols, vals = [], [], []
    nfine = g.size

    def shapes(f):
        lin = [1.0 - f, f]
        if not hermite:
            return lin, lin, None
        return lin, [1.0 - 3 * f ** 2 + 2 * f ** 3, 3 * f ** 2 - 2 * f ** 3], [f * (1.0 - f) ** 2 * Hc, -(1.0 - f) * f ** 2 * Hc]

    sh = [shapes(fr[:, k]) for k in range(3)]
    eps = {(0, 1, 2): 1.0, (1, 2, 0): 1.0, (2, 0, 1): 1.0, (0, 2, 1): -1.0, (2, 1, 0): -1.0, (1, 0, 2): -1.0}
    child = np.arange(nfine)
    for c in range(8):
        bits = [(c >> k) & 1 for k in range(3)]
        gid = ((i0[:, 2] + bits[2]) * (mc + 1) + i0[:, 1] + bits[1]) * (mc + 1) + i0[:, 0] + bits[0]
        w0 = sh[0][1][bits[0]] * sh[1][1][bits[1]] * sh[2][1][bits[2]]
        wl = sh[0][0][bits[0]] * sh[1][0][bits[1]] * sh[2][0][bits[2]]
        keep = (np.abs(w0) > 0) | (np.abs(wl) > 0)
        pos = np.searchsorted(sets[l], gid[keep])
        ok = (pos < sets[l].size)
        ok[ok] &= sets[l][pos[ok]] == gid[keep][ok]
        if not ok.all():
            # a parent the mesh does not touch: only possible with zero weight
            assert np.all(np.abs(w0[keep][~ok]) < 1e-14) and np.all(np.abs(wl[keep][~ok]) < 1e-14)
        kk = np.nonzero(keep)[0][ok]
        pp = pos[ok]
        for i in range(3):
            rows.append(6 * child[kk] + i); cols.append(6 * pp + i); vals.append(w0[kk])
            rows.append(6 * child[kk] + 3 + i); cols.append(6 * pp + 3 + i); vals.append(wl[kk])
        if hermite:
            for j in range(3):
                w1 = sh[j][2][bits[j]]
                for k in range(3):
                    if k != j:
                        w1 = w1 * sh[k][1][bits[k]]
                for (i, k2, jj), sgn in eps.items():
                    if jj == j:
                        rows.append(6 * child[kk] + i); cols.append(6 * pp + 3 + k2); vals.append(sgn * w1[kk])
    T = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * nfine, 6 * sets[l].size))
    T.sum_duplicates()
    return T


def build_nested(hermite_mesh, hermite_lattice):
    sets = node_sets()
    PL = prolongation(levels[-1], hermite_mesh)
    P = [None] * len(levels)
    P[-1] = PL
    for l in range(len(levels) - 2, -1, -1):
        P[l] = (P[l + 1] @ lattice_transfer(l, sets, hermite_lattice)).tocsr()
    A = []
    for Pl in P:
        Al = (Pl.T @ Kf @ Pl).tocsr()
        d = Al.diagonal()
        A.append((Al + sp.diags(np.where(d > 0.0, RIDGE * d, 1.0))).tocsr())
    c = 0
    for l in range(len(levels) - 1):
        if A[l].shape[0] <= cmax:
            c = l
    lu = spla.splu(A[c].tocsc())
    Binv = [block_diag_inv(A[l], 6) if l > c else None for l in range(len(levels))]

    def apply(r):
        z = Spt @ r + P[c] @ lu.solve(P[c].T @ r)
        for l in range(c + 1, len(levels)):
            z += P[l] @ (Binv[l] @ (P[l].T @ r))
        return z
    return apply


for hm, hl, name in ((False, False, "nested, trilinear everywhere (= what runs)"), (True, False, "nested: Hermite from the mesh to the finest lattice, trilinear between lattices"),
                     (True, True, "nested: Hermite from the mesh AND between the lattices")):
    t0 = time.time()
    its, err = pcg(build_nested(hm, hl))
    print(f"{name}: {its} iterations, error {err:.1e} ({time.time()-t0:.1f}s)", flush=True)



# ---- nested Hermite transfers for the APPLICATION, level operators from the DIRECT Hermite interpolation of each level
# (8 nodes per point and level: what the existing Galerkin set-up kernels can form with W = [alpha I, -[sigma]x]) ----------
def build_nested_direct_ops(exact_consistent):
    sets = node_sets()
    PN = [None] * len(levels)
    PN[-1] = prolongation(levels[-1], True)
    for l in range(len(levels) - 2, -1, -1):
        PN[l] = (PN[l + 1] @ lattice_transfer(l, sets, True)).tocsr()
    PD = [prolongation(m, True) for m in levels]
    A = []
    for Pl in PD:
        Al = (Pl.T @ Kf @ Pl).tocsr()
        d = Al.diagonal()
        A.append((Al + sp.diags(np.where(d > 0.0, RIDGE * d, 1.0))).tocsr())
    c = 0
    for l in range(len(levels) - 1):
        if A[l].shape[0] <= cmax:
            c = l
    if exact_consistent:       # coarse operator from the composed prolongation (what a consistent coarse solve needs)
        Ac = (PN[c].T @ Kf @ PN[c]).tocsr(); d = Ac.diagonal()
        lu = spla.splu((Ac + sp.diags(np.where(d > 0.0, RIDGE * d, 1.0))).tocsc())
    else:
        lu = spla.splu(A[c].tocsc())
    Binv = [block_diag_inv(A[l], 6) if l > c else None for l in range(len(levels))]

    def apply(r):
        z = Spt @ r + PN[c] @ lu.solve(PN[c].T @ r)
        for l in range(c + 1, len(levels)):
            z += PN[l] @ (Binv[l] @ (PN[l].T @ r))
        return z
    return apply


for ec in (False, True):
    t0 = time.time()
    its, err = pcg(build_nested_direct_ops(ec))
    print(f"nested Hermite transfers, level operators from the DIRECT Hermite interpolation (coarse operator {'composed' if ec else 'direct'}): {its} iterations, error {err:.1e} ({time.time()-t0:.1f}s)", flush=True)


# ---- non-nested application with direct Hermite P_l on the levels above the coarse solve, nested below? (reference) -------

# ---- what if only the TRANSFERS are Hermite and the level operators stay the trilinear Galerkin ones (no new Galerkin
# kernels on the GPU)?  M^-1 = S + sum_l P_l^H B_l^tri P_l^H^T is still symmetric positive definite -------------------------
def build_mixed():
    sets = node_sets()
    def hierarchy(hm, hl):
        P = [None] * len(levels)
        P[-1] = prolongation(levels[-1], hm)
        for l in range(len(levels) - 2, -1, -1):
            P[l] = (P[l + 1] @ lattice_transfer(l, sets, hl)).tocsr()
        return P
    PH, PT = hierarchy(True, True), hierarchy(False, False)
    A = []
    for Pl in PT:
        Al = (Pl.T @ Kf @ Pl).tocsr()
        d = Al.diagonal()
        A.append((Al + sp.diags(np.where(d > 0.0, RIDGE * d, 1.0))).tocsr())
    c = 0
    for l in range(len(levels) - 1):
        if A[l].shape[0] <= cmax:
            c = l
    lu = spla.splu(A[c].tocsc())
    Binv = [block_diag_inv(A[l], 6) if l > c else None for l in range(len(levels))]

    def apply(r):
        z = Spt @ r + PH[c] @ lu.solve(PH[c].T @ r)
        for l in range(c + 1, len(levels)):
            z += PH[l] @ (Binv[l] @ (PH[l].T @ r))
        return z
    return apply


t0 = time.time()
its, err = pcg(build_mixed())
print(f"Hermite transfers, trilinear Galerkin operators: {its} iterations, error {err:.1e} ({time.time()-t0:.1f}s)", flush=True)
