"""NumPy/SciPy prototype of the shell preconditioner variants (CPU; round 3 investigation of VERDICT item 4).
usage: probe_shell_pc.py [n] [finest] [coarse_unknowns]"""
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
finest = int(sys.argv[2]) if len(sys.argv) > 2 else None
cmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3200
L_ = 25.0
pts, conn = so.scordelis_lo_mesh(n, n)
V = so.ShellSpace(pts, conn)
S = ShellSpace(pts, conn)
t0 = time.time()
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
Kf = (Dm @ K @ Dm + sp.diags(1.0 - mask)).tocsr()         # identity rows / columns on the imposed dofs
b = F * mask
print(f"n={n} dofs={nd} assembled in {time.time()-t0:.1f}s", flush=True)

Lp = lattice_pc(S, finest)
levels, off = Lp["levels"], Lp["level_offsets"]
nl = len(levels)
print("levels", levels, "nodes", np.diff(off))
rows = np.repeat(np.arange(nd), 8)
P = []
for l in range(nl):
    sl = slice(8 * l, 8 * l + 8)
    Pl = sp.csr_matrix((Lp["ell_w"][:, sl].ravel(), (rows, Lp["ell_idx"][:, sl].ravel() - 6 * off[l])), shape=(nd, 6 * (off[l + 1] - off[l])))
    P.append((Dm @ Pl).tocsr())                              # masked: imposed dofs take no correction

# point 3x3 blocks of Kf
def block_diag_inv(A, bs):
    nb = A.shape[0] // bs
    A = A.tocsr()
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
A = [None] * nl
Binv = [None] * nl
for l in range(nl):
    Al = (P[l].T @ Kf @ P[l]).tocsr()
    # lattice unknowns no free dof touches: unit diagonal
    d = Al.diagonal()
    Al = Al + sp.diags((d == 0.0).astype(float))
    A[l] = Al.tocsr()
    Binv[l] = block_diag_inv(A[l], 6)
c = -1
for l in range(nl - 1):
    if A[l].shape[0] <= cmax:
        c = l
print("coarse level", c, levels[c], A[c].shape[0], flush=True)
lu_c = spla.splu(A[c].tocsc())


def pcg(apply_pc, rtol=1e-10, maxit=3000, nmv=None):
    x = np.zeros(nd); r = b.copy(); z = apply_pc(r); p = z.copy(); g = r @ z; g0 = g
    for it in range(1, maxit + 1):
        q = Kf @ p
        a = g / (p @ q)
        x += a * p; r -= a * q
        z = apply_pc(r); g1 = r @ z
        if g1 <= rtol ** 2 * g0:
            return it, x
        p = z + (g1 / g) * p; g = g1
    return maxit, x


def lattice_additive(r, first=None):
    z = np.zeros(nd)
    for l in range(c + 1, nl):
        z += P[l] @ (Binv[l] @ (P[l].T @ r))
    z += P[c] @ lu_c.solve(P[c].T @ r)
    return z


def pc_current(r):
    return Spt @ r + lattice_additive(r)


def lam_max(Sop, iters=30):
    v = np.random.default_rng(0).standard_normal(nd) * mask
    for _ in range(iters):
        w = Sop(Kf @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    return lam


t0 = time.time(); it0, x0 = pcg(pc_current); print(f"current additive: {it0} its ({time.time()-t0:.1f}s)", flush=True)
lm = lam_max(lambda v: Spt @ v)
print("lambda_max(Spt K) =", lm)
om = 1.0 / lm * 1.0

def pc_mult2(r, omega=None):
    """symmetric multiplicative: smoother, lattice correction (additive levels + exact coarse), smoother."""
    w = (omega or (1.3 / lm))
    z = w * (Spt @ r)
    z = z + lattice_additive(r - Kf @ z)
    z = z + w * (Spt @ (r - Kf @ z))
    return z

for fac in (1.0, 1.3, 1.6):
    t0 = time.time(); it, _ = pcg(lambda r: pc_mult2(r, fac / lm)); print(f"mult2 (omega={fac:.1f}/lmax): {it} its, {3} K-applies per it ({time.time()-t0:.1f}s)", flush=True)


def cheb_smoother(r, z0, deg, lmax, lmin_frac=0.25):
    """deg steps of Chebyshev-accelerated point-block Jacobi on K z = r from z0 (targets [lmin_frac lmax, lmax])."""
    a, bq = lmin_frac * lmax, 1.05 * lmax
    theta, delta = 0.5 * (bq + a), 0.5 * (bq - a)
    z = z0.copy()
    res = r - Kf @ z if np.any(z0) else r.copy()
    d = (Spt @ res) / theta
    sigma = theta / delta
    rho_old = 1.0 / sigma
    for k in range(deg):
        z = z + d
        if k == deg - 1:
            break
        res = res - Kf @ d
        rho = 1.0 / (2.0 * sigma - rho_old)
        d = rho * rho_old * d + (2.0 * rho / delta) * (Spt @ res)
        rho_old = rho
    return z


def vcycle(r, l, nu_s=1):
    """multiplicative V-cycle on the lattice hierarchy with Galerkin operators, damped 6x6 block Jacobi smoothing"""
    if l == c:
        return lu_c.solve(r)
    Al, Bl = A[l], Binv[l]
    w = wl[l]
    z = w * (Bl @ r)
    T = Tr[l]                                 # level l <- level l-1 interpolation
    z = z + T @ vcycle(T.T @ (r - Al @ z), l - 1)
    z = z + w * (Bl @ (r - Al @ z))
    return z

# node transfers between lattice levels (6 fields per node)
nn = int(Lp["n_nodes"])
Tp = sp.csr_matrix((Lp["par_vals"], Lp["par_cols"], Lp["par_rowptr"]), shape=(nn, nn))
Tr = [None] * nl
I6 = sp.identity(6, format="csr")
for l in range(1, nl):
    Tn = Tp[off[l]:off[l + 1], off[l - 1]:off[l]]
    Tr[l] = sp.kron(Tn, I6, format="csr")
    # consistency: P_{l-1} = P_l T_l
    if l == nl - 1:
        print("nested check", abs(P[l] @ Tr[l] - P[l - 1]).max())
wl = [None] * nl
for l in range(c + 1, nl):
    v = np.random.default_rng(1).standard_normal(A[l].shape[0])
    for _ in range(30):
        w = Binv[l] @ (A[l] @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    wl[l] = 1.3 / lam
    print("level", levels[l], "lmax(B^-1 A)", lam)

def pc_vcycle(r, fac=1.3):
    w = fac / lm
    z = w * (Spt @ r)
    rr = r - Kf @ z
    z = z + P[nl - 1] @ vcycle(P[nl - 1].T @ rr, nl - 1)
    z = z + w * (Spt @ (r - Kf @ z))
    return z

t0 = time.time(); it, _ = pcg(pc_vcycle); print(f"V-cycle (Galerkin levels, block-Jacobi smoothing): {it} its ({time.time()-t0:.1f}s)", flush=True)

def pc_vcycle_cheb(r, deg=2):
    z = cheb_smoother(r, np.zeros(nd), deg, lm)
    rr = r - Kf @ z
    z = z + P[nl - 1] @ vcycle(P[nl - 1].T @ rr, nl - 1)
    # symmetric post-smoothing: same polynomial applied to the new residual
    z = z + cheb_smoother(r - Kf @ z, np.zeros(nd), deg, lm)
    return z

for deg in (2, 3):
    t0 = time.time(); it, _ = pcg(lambda r: pc_vcycle_cheb(r, deg)); print(f"V-cycle + Chebyshev({deg}) fine smoother: {it} its, {2*deg+1} K-applies per it ({time.time()-t0:.1f}s)", flush=True)

# ---- limits: exact solve on the FINEST lattice level
luF = spla.splu(A[nl - 1].tocsc())
def pc_twogrid_add(r):
    return Spt @ r + P[nl - 1] @ luF.solve(P[nl - 1].T @ r)
def pc_twogrid_mult(r, fac=1.3):
    w = fac / lm
    z = w * (Spt @ r)
    z = z + P[nl - 1] @ luF.solve(P[nl - 1].T @ (r - Kf @ z))
    z = z + w * (Spt @ (r - Kf @ z))
    return z
t0 = time.time(); it, _ = pcg(pc_twogrid_add); print(f"two-grid additive, exact finest lattice ({A[nl-1].shape[0]} unknowns): {it} its", flush=True)
t0 = time.time(); it, _ = pcg(pc_twogrid_mult); print(f"two-grid multiplicative, exact finest lattice: {it} its", flush=True)
for deg in (2, 3):
    def pc_tg_cheb(r, deg=deg):
        z = cheb_smoother(r, np.zeros(nd), deg, lm)
        z = z + P[nl - 1] @ luF.solve(P[nl - 1].T @ (r - Kf @ z))
        z = z + cheb_smoother(r - Kf @ z, np.zeros(nd), deg, lm)
        return z
    it, _ = pcg(pc_tg_cheb); print(f"two-grid + Chebyshev({deg}): {it} its", flush=True)

# V-cycle with more smoothing on the lattice levels (cheap: the levels are small)
def vcycle_k(r, l, ks):
    if l == c:
        return lu_c.solve(r)
    Al, Bl = A[l], Binv[l]
    w = wl[l]
    z = np.zeros_like(r)
    for _ in range(ks):
        z = z + w * (Bl @ (r - Al @ z))
    T = Tr[l]
    z = z + T @ vcycle_k(T.T @ (r - Al @ z), l - 1, ks)
    for _ in range(ks):
        z = z + w * (Bl @ (r - Al @ z))
    return z
for ks in (2, 4):
    def pc_vk(r, ks=ks, fac=1.3):
        w = fac / lm
        z = w * (Spt @ r)
        z = z + P[nl - 1] @ vcycle_k(P[nl - 1].T @ (r - Kf @ z), nl - 1, ks)
        z = z + w * (Spt @ (r - Kf @ z))
        return z
    it, _ = pcg(pc_vk); print(f"V-cycle with {ks} smoothing steps per lattice level: {it} its", flush=True)

# ---- smoothed prolongation (smoothed-aggregation flavour): Pt = (I - w S K) P, Galerkin operators from Pt
print("--- smoothed prolongation", flush=True)
for fac in (0.66, 1.0, 1.33):
    wS = fac / lm
    SK = (Spt @ Kf).tocsr()
    PtF = (P[nl - 1] - wS * (SK @ P[nl - 1])).tocsr()
    Pt = [None] * nl
    Pt[nl - 1] = PtF
    for l in range(nl - 2, -1, -1):
        Pt[l] = (Pt[l + 1] @ Tr[l + 1]).tocsr()
    At, Bt = [None] * nl, [None] * nl
    for l in range(c, nl):
        Al = (Pt[l].T @ Kf @ Pt[l]).tocsr()
        d = Al.diagonal()
        Al = (Al + sp.diags((d == 0.0).astype(float))).tocsr()
        At[l] = Al
        Bt[l] = block_diag_inv(Al, 6)
    lut = spla.splu(At[c].tocsc())
    def pc_sa_add(r):
        z = Spt @ r
        for l in range(c + 1, nl):
            z = z + Pt[l] @ (Bt[l] @ (Pt[l].T @ r))
        return z + Pt[c] @ lut.solve(Pt[c].T @ r)
    it, _ = pcg(pc_sa_add); print(f"smoothed P (omega={fac:.2f}/lmax), additive: {it} its (3 K-applies per it)", flush=True)
    wls = {}
    for l in range(c + 1, nl):
        v = np.random.default_rng(1).standard_normal(At[l].shape[0])
        for _ in range(30):
            w = Bt[l] @ (At[l] @ v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
        wls[l] = 1.3 / lam
    def vc(r, l):
        if l == c:
            return lut.solve(r)
        z = wls[l] * (Bt[l] @ r)
        T = Tr[l]
        z = z + T @ vc(T.T @ (r - At[l] @ z), l - 1)
        return z + wls[l] * (Bt[l] @ (r - At[l] @ z))
    def pc_sa_v(r, f2=1.3):
        w = f2 / lm
        z = w * (Spt @ r)
        z = z + PtF @ vc(PtF.T @ (r - Kf @ z), nl - 1)
        return z + w * (Spt @ (r - Kf @ z))
    it, _ = pcg(pc_sa_v); print(f"smoothed P (omega={fac:.2f}/lmax), V-cycle: {it} its (5 K-applies per it)", flush=True)

# ---- deflation with Ritz vectors harvested from the forward solve (recycling for the adjoint right-hand side)
print("--- deflation / recycling", flush=True)
def pcg_store(apply_pc, rhs, rtol=1e-10, maxit=3000):
    x = np.zeros(nd); r = rhs.copy(); z = apply_pc(r); p = z.copy(); g = r @ z; g0 = g
    Z, al, be = [z / np.sqrt(g)], [], []
    for it in range(1, maxit + 1):
        q = Kf @ p
        a = g / (p @ q)
        x += a * p; r -= a * q
        z = apply_pc(r); g1 = r @ z
        al.append(a)
        if g1 <= rtol ** 2 * g0:
            return it, x, Z, al, be
        be.append(g1 / g)
        Z.append(z / np.sqrt(g1) * (-1) ** it)
        p = z + (g1 / g) * p; g = g1
    return maxit, x, Z, al, be

it, x, Z, al, be = pcg_store(pc_current, b)
m = len(al)
T = np.zeros((m, m))
for j in range(m):
    T[j, j] = 1.0 / al[j] + (be[j - 1] / al[j - 1] if j > 0 else 0.0)
    if j + 1 < m:
        T[j, j + 1] = T[j + 1, j] = np.sqrt(be[j]) / al[j]
ev, V = np.linalg.eigh(T)
print("forward solve", it, "its; extreme Ritz values of M^-1 K:", ev[:6], "...", ev[-3:], " cond ~", ev[-1] / ev[0])
Zm = np.stack(Z[:m], axis=1)
w0 = spla.splu(Kf.tocsc()).solve(b)
rhs2 = so.compliance_du(V_ := V, w0) if False else so.compliance_du(so.ShellSpace(pts, conn), w0) * mask
it_plain, _ = pcg(pc_current) if False else (None, None)
def pcg_rhs(apply_pc, rhs, rtol=1e-10, maxit=3000, W=None):
    if W is not None:
        KW = Kf @ W
        E = W.T @ KW
        Einv = np.linalg.inv(E)
        x = W @ (Einv @ (W.T @ rhs))
        r = rhs - Kf @ x
    else:
        x = np.zeros(nd); r = rhs.copy()
    def proj(v):                       # v - W E^-1 (KW)^T v : K-orthogonal to W
        return v - W @ (Einv @ (KW.T @ v)) if W is not None else v
    z = proj(apply_pc(r)); p = z.copy(); g = r @ z; g0 = rhs @ apply_pc(rhs)
    for it in range(1, maxit + 1):
        q = Kf @ p
        a = g / (p @ q)
        x += a * p; r -= a * q
        z = proj(apply_pc(r)); g1 = r @ z
        if g1 <= rtol ** 2 * g0:
            return it
        p = z + (g1 / g) * p; g = g1
    return maxit
print("adjoint rhs, no deflation:", pcg_rhs(pc_current, rhs2), "its")
for k in (4, 8, 16, 32):
    W = Zm @ V[:, :k]                  # Ritz vectors of the k smallest Ritz values
    print(f"adjoint rhs, deflating {k} Ritz vectors: {pcg_rhs(pc_current, rhs2, W=W)} its")
