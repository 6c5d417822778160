"""NumPy prototype: does the adjoint solve of a cycle profit from the forward solve's search directions?
Seed projection (Erhel / Guyomarc'h): lam0 = sum_k p_k (p_k . c) / (p_k . A p_k) over the stored directions, then PCG on
the remainder.  Poisson (BPX-PCG, oracle operators) and the shell (additive lattice preconditioner of probe_shell_pc).
usage: probe_recycle.py [n_poisson] [n_shell]"""
import os
import sys
import math

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import femo_oracle as fo
from oracle import bpx_oracle as bo


def pcg_store(A, b, apply_pc, rtol, x0=None, store=None, tol_ref=None, max_it=5000):
    """PCG, stopping on sqrt(r.z) <= rtol sqrt(b.M^-1 b) (tol_ref: that reference, for solves that start from a guess)."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    z = apply_pc(r)
    p = z.copy()
    rz = float(r @ z)
    ref = tol_ref if tol_ref is not None else rz
    tol2 = rtol * rtol * ref
    it = 0
    if rz <= tol2:
        return x, 0
    while it < max_it:
        q = A @ p
        d = float(p @ q)
        if store is not None:
            store.append((p.copy(), d))
        a = rz / d
        x += a * p
        r -= a * q
        it += 1
        z = apply_pc(r)
        rz1 = float(r @ z)
        if rz1 <= tol2:
            break
        p = z + (rz1 / rz) * p
        rz = rz1
    return x, it


def seed_guess(store, c):
    x = np.zeros_like(c)
    for p, d in store:
        x += p * (float(p @ c) / d)
    return x


def poisson(n):
    m = fo.unit_cube_mesh(n)
    bd = fo.boundary_vertices_box(m.x)
    K = fo.stiffness(m).tocsr()
    A = fo.eliminate_bc(K, bd).tocsr()
    xc = fo.centroids(m)
    f = np.prod(np.sin(np.pi * xc), axis=1) * (0.5 + 0.4 * np.cos(2 * np.pi * xc[:, 0]) * np.cos(np.pi * xc[:, 1]) + 0.3 * xc[:, 2])
    b = fo.load_vector(m, f)
    b[bd] = 0.0
    pinned = np.zeros(m.n_vert, bool); pinned[bd] = True
    M = bo.BPX(m.x, A.diagonal(), pinned)
    store = []
    u, it_f = pcg_store(A, b, M.apply, 1e-11, store=store)
    Mm = fo.mass_matrix(m) if hasattr(fo, "mass_matrix") else None
    ud = fo.u_target(m.x)
    c = (Mm @ (u - ud)) if Mm is not None else (u - ud)
    c[bd] = 0.0
    ref = float(c @ M.apply(c))
    _, it_a0 = pcg_store(A, c, M.apply, 1e-11)
    lam0 = seed_guess(store, c)
    r0 = c - A @ lam0
    print(f"Poisson n={n}: forward {it_f} its; adjoint from zero {it_a0} its; seed guess leaves sqrt(r.M^-1 r) = "
          f"{math.sqrt(float(r0 @ M.apply(r0)) / ref):.2e} of the rhs", flush=True)
    _, it_a1 = pcg_store(A, c, M.apply, 1e-11, x0=lam0, tol_ref=ref)
    print(f"   adjoint from the seed guess: {it_a1} its (same absolute threshold)", flush=True)


def shell(n):
    from femo_amd.fea.shell import ShellSpace, lattice_pc
    from oracle import shell_oracle as so
    pts, conn = so.scordelis_lo_mesh(n, n)
    V = so.ShellSpace(pts, conn)
    S = ShellSpace(pts, conn)
    K = so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, 0.0)).tocsr()
    F = so.load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1)))
    ux, vx, L_ = V.unode_x, V.x, 25.0
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

    def bdinv(A, bs):
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

    Spt = bdinv(Kf, 3)
    A = []
    for l in range(nl):
        Al = (P[l].T @ Kf @ P[l]).tocsr()
        A.append((Al + sp.diags((Al.diagonal() == 0.0).astype(float))).tocsr())
    c_ = 0
    for l in range(nl - 1):
        if A[l].shape[0] <= 3200:
            c_ = l
    lu = spla.splu(A[c_].tocsc())
    Binv = [bdinv(A[l], 6) if l > c_ else None for l in range(nl)]

    def apply_pc(r):
        z = Spt @ r + P[c_] @ lu.solve(P[c_].T @ r)
        for l in range(c_ + 1, nl):
            z += P[l] @ (Binv[l] @ (P[l].T @ r))
        return z

    store = []
    w, it_f = pcg_store(Kf, b, apply_pc, 1e-10, store=store)
    cadj = so.compliance_du(V, w) * mask
    ref = float(cadj @ apply_pc(cadj))
    _, it_a0 = pcg_store(Kf, cadj, apply_pc, 1e-10)
    lam0 = seed_guess(store, cadj)
    r0 = cadj - Kf @ lam0
    print(f"shell {n} x {n}: forward {it_f} its; adjoint from zero {it_a0} its; seed guess leaves {math.sqrt(float(r0 @ apply_pc(r0)) / ref):.2e} of the rhs", flush=True)
    _, it_a1 = pcg_store(Kf, cadj, apply_pc, 1e-10, x0=lam0, tol_ref=ref)
    print(f"   adjoint from the seed guess: {it_a1} its", flush=True)


if __name__ == "__main__":
    poisson(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    shell(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
