"""does a real multigrid solve on the auxiliary lattice (Xu's auxiliary-space form,
M^-1 = D^-1 + P B P^T with B ~ (lattice operator)^-1) bring the PCG count from 28 to <= 18 at rtol 1e-11 without a
second fine-level SpMV per iteration?  Oracle-level experiment (NumPy / SciPy on the CPU), gate before any device code.

Variants of B on the finest lattice (every coarser level only matters through how well B inverts the finest operator):
  bpx      the additive operator of oracle/bpx_oracle.py (the current design)               -- reference count
  exactG   B = (P^T A P)^-1 restricted to the kept nodes (Galerkin, exact)                  -- best any V-cycle can do
  exactQ   B = (Q1 lattice Laplacian on the kept nodes)^-1 (exact)                          -- best a stencil V-cycle can do
  vcycQ    symmetric V(1,1) cycle, damped Jacobi, on the Q1 lattice Laplacian chain         -- what the device would run
each with the fine-level Jacobi term weighted by omega (additive, so no extra SpMV).
"""
import sys
import math
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from oracle import bpx_oracle as bo
from oracle import femo_oracle as fo


def system(m, seed=0):
    bd = fo.boundary_vertices_box(m.x)
    A = fo.eliminate_bc(fo.stiffness(m), bd).tocsr()
    rng = np.random.default_rng(seed)
    b = fo.load_vector(m, 1.0 + rng.random(m.n_cell))
    b[bd] = 0.0
    pinned = np.zeros(m.n_vert, bool)
    pinned[bd] = True
    return A, b, pinned


def q1_laplacian(bins, H, dim):
    """Q1 stiffness on the lattice (bins per axis, spacing H, x fastest)."""
    def k1(n):   # 1-D stiffness / mass of linear elements with unit spacing
        K = sp.diags([-np.ones(n), np.r_[1.0, 2.0 * np.ones(n - 1), 1.0], -np.ones(n)], [-1, 0, 1])
        Mm = sp.diags([np.ones(n) / 6, np.r_[1.0 / 3, 2.0 / 3 * np.ones(n - 1), 1.0 / 3], np.ones(n) / 6], [-1, 0, 1])
        return K.tocsr(), Mm.tocsr()
    KM = [k1(int(n)) for n in bins]
    out = None
    for a in range(dim):
        term = None
        for k in range(dim):       # kron with x fastest: later axes on the left
            f = KM[k][0] if k == a else KM[k][1]
            term = f if term is None else sp.kron(f, term, format="csr")
        out = term if out is None else out + term
    return (out * H ** (dim - 2)).tocsr()


class Aux:
    def __init__(self, M: bo.BPX, A, kind, omega=1.0, nu=1, wj=0.8, sigma=1.0, cycles=1):
        self.M, self.kind, self.omega, self.sigma = M, kind, omega, sigma
        self.dinv = M.dinv
        L = M.levels - 1
        self.keep = [c != 0.0 for c in M.coef]
        P = M.P
        self.P = P
        self.nu, self.wj, self.cycles = nu, wj, cycles
        if kind == "exactG":
            G = (P.T @ A @ P).tocsr()
            k = np.nonzero(self.keep[L] & (G.diagonal() > 0))[0]
            self.k = k
            self.lu = spla.splu(G[k][:, k].tocsc())
        elif kind in ("exactQ", "vcycQ"):
            self.Q = []
            for l in range(L + 1):
                Ql = q1_laplacian(M.bins[l], M.H[l], M.dim)
                kp = self.keep[l].astype(float)
                D = sp.diags(kp)
                Ql = (D @ Ql @ D + sp.diags(1.0 - kp)).tocsr()
                self.Q.append(Ql)
            if kind == "exactQ":
                self.lu = spla.splu(self.Q[L].tocsc())
            else:
                self.lu0 = spla.splu(self.Q[0].tocsc())

    def vcycle(self, l, g):
        Q = self.Q[l]
        kp = self.keep[l]
        if l == 0:
            return np.where(kp, self.lu0.solve(g * kp), 0.0)
        dinv = self.wj / Q.diagonal()
        e = np.zeros_like(g)
        for _ in range(self.nu):
            e = e + dinv * (g * kp - Q @ e) * kp
        I = self.M.I[l - 1]
        rc = I.T @ ((g - Q @ e) * kp)
        e = e + (I @ self.vcycle(l - 1, rc * self.keep[l - 1])) * kp
        for _ in range(self.nu):
            e = e + dinv * (g * kp - Q @ e) * kp
        return e

    def apply(self, r):
        M = self.M
        if self.kind == "bpx":
            return M.apply(r)
        L = M.levels - 1
        g = self.P.T @ r
        if self.kind == "exactG":
            e = np.zeros_like(g)
            e[self.k] = self.lu.solve(g[self.k])
        elif self.kind == "exactQ":
            e = self.lu.solve(g * self.keep[L]) * self.keep[L]
        else:
            e = self.vcycle(L, g * self.keep[L])
        return self.omega * self.dinv * r + self.sigma * (self.P @ e)


def pcg(A, b, B, rtol=1e-11, max_it=500):
    x = np.zeros_like(b)
    r = b.copy()
    z = B.apply(r)
    p = z.copy()
    rz = float(r @ z)
    tol2 = rtol * rtol * rz
    it = 0
    while it < max_it:
        q = A @ p
        alpha = rz / float(p @ q)
        x += alpha * p
        r -= alpha * q
        it += 1
        z = B.apply(r)
        rzn = float(r @ z)
        if rzn <= tol2:
            break
        p = z + (rzn / rz) * p
        rz = rzn
    return x, it


def kappa(A, B, n):
    """extreme eigenvalues of B A by Lanczos on the symmetrised operator (small meshes)."""
    op = spla.LinearOperator((n, n), matvec=lambda v: B.apply(A @ v))
    lmax = spla.eigs(op, k=1, which="LM", return_eigenvectors=False, tol=1e-4)[0].real
    lmin = spla.eigs(op, k=1, which="SM", return_eigenvectors=False, tol=1e-3, maxiter=5000)[0].real if n < 6000 else float("nan")
    return lmin, lmax


if __name__ == "__main__":
    cases = [(3, 24, 0.0), (3, 40, 0.0), (3, 32, 0.2), (2, 128, 0.0)]
    if len(sys.argv) > 1:
        cases = [(3, int(sys.argv[1]), 0.0)]
    for d, n, jit in cases:
        m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
        A, b, pinned = system(m, seed=n)
        M = bo.BPX(m.x, A.diagonal(), pinned)
        xref = spla.spsolve(A.tocsc(), b) if m.n_vert < 80000 else None
        print(f"--- d={d} n={n} jitter={jit}: {m.n_vert} vertices, lattice {list(M.bins[-1])}, {M.levels} levels")
        for kind, kw in [("bpx", {}),
                         ("exactG", dict(omega=1.0)), ("exactG", dict(omega=0.7)), ("exactG", dict(omega=0.5)),
                         ("exactQ", dict(omega=1.0)), ("exactQ", dict(omega=0.7)), ("exactQ", dict(omega=0.5, sigma=1.0)),
                         ("exactQ", dict(omega=0.7, sigma=1.5)), ("exactQ", dict(omega=0.7, sigma=0.7)),
                         ("vcycQ", dict(omega=0.7, nu=1)), ("vcycQ", dict(omega=0.7, nu=2)),
                         ]:
            B = Aux(M, A, kind, **kw)
            x, it = pcg(A, b, B)
            err = np.abs(x - xref).max() / np.abs(xref).max() if xref is not None else float("nan")
            print(f"  {kind:7s} {str(kw):44s} its {it:3d}  err {err:.1e}", flush=True)
