"""CPU oracle for femo's PDE-residual hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the *checker*.  Nothing under
``femo_amd/`` imports it; the product path fails loudly when the HIP
extension is missing.

PARITY UNPINNED for the Poisson forms (pinned only for the beam, see below).  The arithmetic
of the reference lives in un-vendored
third-party packages (dolfinx 0.5.1 / UFL / FFCx / PETSc / MUMPS, pinned only by
``README.md:19`` of the reference) that cannot be imported in the build
container, and the reference ships no tests, golden vectors or fixtures for
this path.  This file therefore *restates* the published
algorithm of those packages for the forms the reference's examples define, in
NumPy/SciPy, and is pinned against closed-form known answers only
(``tests/test_oracle_*.py``): exact P1 element matrices, the DST-exact discrete
Poisson solve on structured grids, finite differences of the functional, and
the analytic optimal-control pair of ``examples/poisson_opt``.
The one golden vector the reference tree holds for this path -- the 50 OpenMDAO-optimal
thicknesses of ``examples/beam_thickness_opt/run_thickness_opt_cantilever_beam.py:252-261`` --
pins the Euler-Bernoulli beam part of this oracle (``BEAM_THICK_REF``, reproduced to 4e-7 by
re-running SLSQP with the oracle's adjoint gradients in tests/test_oracle.py).

Citations are relative to /root/reference/.  "[ext]" marks behaviour of a
third-party package restated from its documentation.

Conventions
-----------
* P1 (CG1) simplices, ``tdim`` in {2, 3}; the state ``u`` lives on vertices
  (shape ``(n_vert,)``), the source ``f`` is DG0 (shape ``(n_cell,)``), as in
  ``examples/poisson_opt/run_poisson_opt.py:98-105``.
* All matrices are ``scipy.sparse.csr_matrix`` with sorted indices; explicit
  zeros created by Dirichlet elimination are kept so patterns compare equal.
* DOF numbering = vertex numbering of the mesh (dolfinx renumbers DOFs [ext];
  every fixture is therefore keyed by this module's own mesh generators, which
  the product re-implements in ``femo_amd/fea/mesh.py`` and the tests compare
  by coordinates).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ALPHA_POISSON = 1e-6  # run_poisson_opt.py:28,112


# --------------------------------------------------------------------------
# meshes  (femo/fea/utils_dolfinx.py:136-140 createUnitSquareMesh -> dolfinx
# create_unit_square, DiagonalType.right [ext]; the cube is the 3-D analogue
# dolfinx create_unit_cube, 6 tetrahedra per hexahedron around the main
# diagonal [ext])
# --------------------------------------------------------------------------
@dataclass
class OMesh:
    tdim: int
    x: np.ndarray      # (n_vert, tdim) float64
    conn: np.ndarray   # (n_cell, tdim+1) int32
    n: int = 0         # structured resolution (0 = unstructured)

    @property
    def n_vert(self) -> int:
        return self.x.shape[0]

    @property
    def n_cell(self) -> int:
        return self.conn.shape[0]


def _jitter(x: np.ndarray, n: int, interior: np.ndarray, jitter: float, seed: int) -> np.ndarray:
    if jitter == 0.0:
        return x
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1.0, 1.0, size=x.shape) * (jitter / n)
    x = x.copy()
    x[interior] += d[interior]
    return x


def unit_square_mesh(n: int, jitter: float = 0.0, seed: int = 20240807) -> OMesh:
    """(n+1)^2 vertices, lexicographic (x fastest); 2 n^2 right-diagonal triangles."""
    g = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(g, g, indexing="xy")           # row = y, col = x
    x = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    v0 = (j * (n + 1) + i).ravel()
    v1 = v0 + 1
    v2 = v0 + (n + 1)
    v3 = v2 + 1
    conn = np.empty((2 * n * n, 3), dtype=np.int32)
    conn[0::2] = np.stack([v0, v1, v3], axis=1)
    conn[1::2] = np.stack([v0, v3, v2], axis=1)
    on_b = boundary_vertices_box(x)
    interior = np.ones(x.shape[0], bool)
    interior[on_b] = False
    return OMesh(2, _jitter(x, n, interior, jitter, seed), conn, n)


def unit_cube_mesh(n: int, jitter: float = 0.0, seed: int = 20240807) -> OMesh:
    """(n+1)^3 vertices, lexicographic (x fastest, z slowest); 6 n^3 Kuhn tetrahedra."""
    g = np.linspace(0.0, 1.0, n + 1)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    np1 = n + 1
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = (k * np1 * np1 + j * np1 + i).ravel()
    off = lambda dx, dy, dz: base + dx + dy * np1 + dz * np1 * np1
    # Kuhn triangulation: one tet per permutation of the axes, all sharing the
    # diagonal (0,0,0)-(1,1,1).
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    conn = np.empty((6 * n ** 3, 4), dtype=np.int32)
    for t, p in enumerate(perms):
        d = [0, 0, 0]
        verts = [off(0, 0, 0)]
        for ax in p:
            d[ax] = 1
            verts.append(off(*d))
        conn[t::6] = np.stack(verts, axis=1)
    on_b = boundary_vertices_box(x)
    interior = np.ones(x.shape[0], bool)
    interior[on_b] = False
    return OMesh(3, _jitter(x, n, interior, jitter, seed), conn, n)


def boundary_vertices_box(x: np.ndarray, atol: float = 1e-6) -> np.ndarray:
    """Vertices with any coordinate on {0, 1} (run_poisson_opt.py:124-133)."""
    on = np.zeros(x.shape[0], bool)
    for k in range(x.shape[1]):
        on |= np.isclose(x[:, k], 0.0, atol=atol) | np.isclose(x[:, k], 1.0, atol=atol)
    return np.nonzero(on)[0].astype(np.int32)


# --------------------------------------------------------------------------
# P1 element geometry
# --------------------------------------------------------------------------
def cell_geometry(mesh: OMesh) -> Tuple[np.ndarray, np.ndarray]:
    """Volumes (n_cell,) and constant basis gradients (n_cell, tdim+1, tdim)."""
    d = mesh.tdim
    X = mesh.x[mesh.conn]                                # (nc, d+1, d)
    E = X[:, 1:, :] - X[:, :1, :]                        # rows = edge vectors
    det = np.linalg.det(E)
    vol = np.abs(det) / math.factorial(d)
    Einv = np.linalg.inv(E)                              # columns = grad phi_1..d
    g = np.empty((mesh.n_cell, d + 1, d))
    g[:, 1:, :] = np.swapaxes(Einv, 1, 2)
    g[:, 0, :] = -g[:, 1:, :].sum(axis=1)
    return vol, g


def _scatter_matrix(mesh: OMesh, Ke: np.ndarray, n_rows: int, n_cols: int,
                    rows: np.ndarray, cols: np.ndarray) -> sp.csr_matrix:
    A = sp.coo_matrix((Ke.ravel(), (rows.ravel(), cols.ravel())), shape=(n_rows, n_cols)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


# --------------------------------------------------------------------------
# forms of examples/poisson_opt  (a6, a10, a14, a15 in SURVEY.md section 8(a))
# --------------------------------------------------------------------------
def stiffness(mesh: OMesh) -> sp.csr_matrix:
    """dR/du of  inner(grad u, grad v) dx  (run_poisson_opt.py:37), no BCs
    (state_model.py:132 -> utils_dolfinx.py:181-187 assembleMatrix with bcs=[])."""
    vol, g = cell_geometry(mesh)
    Ke = vol[:, None, None] * np.einsum("cad,cbd->cab", g, g)
    r = np.repeat(mesh.conn[:, :, None], mesh.tdim + 1, axis=2)
    c = np.repeat(mesh.conn[:, None, :], mesh.tdim + 1, axis=1)
    return _scatter_matrix(mesh, Ke, mesh.n_vert, mesh.n_vert, r, c)


def residual(mesh: OMesh, u: np.ndarray, f: np.ndarray) -> np.ndarray:
    """R_i = int grad u . grad phi_i - int f phi_i  (run_poisson_opt.py:32-38),
    assembled WITHOUT any BC treatment (state_model.py:85 -> utils:175-179)."""
    vol, g = cell_geometry(mesh)
    ue = u[mesh.conn]                                   # (nc, d+1)
    gu = np.einsum("cbd,cb->cd", g, ue)                 # grad u per cell
    Re = vol[:, None] * np.einsum("cad,cd->ca", g, gu) - (f * vol / (mesh.tdim + 1))[:, None]
    R = np.zeros(mesh.n_vert)
    np.add.at(R, mesh.conn.ravel(), Re.ravel())
    return R


def dRdf(mesh: OMesh) -> sp.csr_matrix:
    """dR/df for DG0 f: -(int_c phi_a) = -|T_c|/(d+1) at (v_a, c); no BCs
    (state_model.py:136-146)."""
    vol, _ = cell_geometry(mesh)
    d1 = mesh.tdim + 1
    vals = np.repeat(-vol / d1, d1)
    rows = mesh.conn.ravel()
    cols = np.repeat(np.arange(mesh.n_cell), d1)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(mesh.n_vert, mesh.n_cell)).tocsr()
    A.sort_indices()
    return A


def mass_apply_cell(mesh: OMesh, e: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-cell  int e^2  and the element vectors  M_e e_e  for P1 e."""
    vol, _ = cell_geometry(mesh)
    d = mesh.tdim
    ee = e[mesh.conn]
    s = ee.sum(axis=1)
    c = vol / ((d + 1) * (d + 2))
    int_e2 = c * ((ee ** 2).sum(axis=1) + s ** 2)
    Me = c[:, None] * (ee + s[:, None])
    return int_e2, Me


def functional(mesh: OMesh, u: np.ndarray, f: np.ndarray, u_d: np.ndarray,
               alpha: float = ALPHA_POISSON) -> float:
    """J = 1/2 int (u-u_d)^2 + alpha/2 int f^2  (run_poisson_opt.py:74-76);
    u_d is the CG1 interpolant (fea_dolfinx.py:163-167)."""
    vol, _ = cell_geometry(mesh)
    int_e2, _ = mass_apply_cell(mesh, u - u_d)
    return 0.5 * int_e2.sum() + 0.5 * alpha * (f * f * vol).sum()


def functional_du(mesh: OMesh, u: np.ndarray, u_d: np.ndarray) -> np.ndarray:
    """dJ/du = M (u - u_d)   (output_model.py:82-87, dim=1)."""
    _, Me = mass_apply_cell(mesh, u - u_d)
    out = np.zeros(mesh.n_vert)
    np.add.at(out, mesh.conn.ravel(), Me.ravel())
    return out


def functional_df(mesh: OMesh, f: np.ndarray, alpha: float = ALPHA_POISSON) -> np.ndarray:
    """dJ/df_c = alpha f_c |T_c|."""
    vol, _ = cell_geometry(mesh)
    return alpha * f * vol


def f_star(xc: np.ndarray, alpha: float = ALPHA_POISSON) -> np.ndarray:
    """run_poisson_opt.py:78-84 (d-dimensional product form)."""
    return np.prod(np.sin(np.pi * xc), axis=1) / (1.0 + alpha * 4.0 * np.pi ** 4)


def u_target(x: np.ndarray) -> np.ndarray:
    """run_poisson_opt.py:86-92; 1/(d pi^2) prod sin(pi x_k)."""
    d = x.shape[1]
    return np.prod(np.sin(np.pi * x), axis=1) / (d * np.pi ** 2)


def centroids(mesh: OMesh) -> np.ndarray:
    return mesh.x[mesh.conn].mean(axis=1)


# --------------------------------------------------------------------------
# Dirichlet algebra  (a16: utils_dolfinx.py:189-202; dolfinx assemble_matrix
# with bcs zeroes BC rows+cols and puts 1 on the diagonal [ext])
# --------------------------------------------------------------------------
def eliminate_bc(K: sp.csr_matrix, bc_dofs: np.ndarray) -> sp.csr_matrix:
    """Same pattern as K; rows/cols of bc_dofs zeroed, diagonal 1."""
    A = K.tocsr(copy=True)
    A.sort_indices()
    isbc = np.zeros(A.shape[0], bool)
    isbc[bc_dofs] = True
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    kill = isbc[rows] | isbc[A.indices]
    A.data[kill] = 0.0
    A.data[kill & (rows == A.indices)] = 1.0
    return A


def newton_rhs(K: sp.csr_matrix, F: np.ndarray, u: np.ndarray,
               bc_dofs: np.ndarray, bc_vals: np.ndarray) -> np.ndarray:
    """dolfinx NonlinearProblem.F [ext] as driven from utils_dolfinx.py:431:
    b = F; apply_lifting(b,[a],[bcs],x0=[u],scale=-1): b -= -1 * K[:,bc](g - u);
    set_bc(b,bcs,u,-1): b[bc] = -(g - u)[bc]."""
    w = np.zeros_like(u)
    w[bc_dofs] = bc_vals - u[bc_dofs]
    b = F + K @ w
    b[bc_dofs] = u[bc_dofs] - bc_vals
    return b


@dataclass
class SolveInfo:
    newton_its: int = 0
    residual_norms: list = field(default_factory=list)


def newton_solve(mesh: OMesh, f: np.ndarray, u0: np.ndarray, bc_dofs: np.ndarray,
                 bc_vals: np.ndarray, max_it: int = 3, initialize: bool = False
                 ) -> Tuple[np.ndarray, SolveInfo]:
    """utils_dolfinx.py:419-449 'Newton': atol 1e-50, rtol 1e-30, max_it 3 =>
    always exactly 3 iterations of  assemble F (+lifting) / assemble J(bcs) /
    LU / x -= dx  [ext dolfinx.nls.petsc.NewtonSolver, relaxation 1]."""
    u = u0.copy()
    if initialize:
        u[:] = 0.1                                     # utils_dolfinx.py:433-435
    K = stiffness(mesh)
    info = SolveInfo()
    for _ in range(max_it):
        F = residual(mesh, u, f)
        b = newton_rhs(K, F, u, bc_dofs, bc_vals)
        A = eliminate_bc(K, bc_dofs)
        dx = spla.splu(A.tocsc()).solve(b)
        u -= dx
        info.newton_its += 1
        info.residual_norms.append(float(np.linalg.norm(b)))
    return u, info


# --------------------------------------------------------------------------
# linearisation + adjoint  (a10-a13)
# --------------------------------------------------------------------------
@dataclass
class Linearization:
    dRdu: sp.csr_matrix     # no BCs          state_model.py:132
    dRdf: sp.csr_matrix     # no BCs          state_model.py:136-146
    A: sp.csr_matrix        # BCs eliminated  state_model.py:149-151


def linearize(mesh: OMesh, bc_dofs: np.ndarray) -> Linearization:
    K = stiffness(mesh)
    return Linearization(K, dRdf(mesh), eliminate_bc(K, bc_dofs))


def solve_linear_bwd(A: sp.csr_matrix, du_seed: np.ndarray) -> np.ndarray:
    """fea_dolfinx.py:208-222 with ksp=None: dR = (A^T)^{-1} du."""
    return spla.splu(A.T.tocsc()).solve(du_seed)


def solve_linear_fwd_reference(A: sp.csr_matrix, dR_seed: np.ndarray) -> np.ndarray:
    """fea_dolfinx.py:192-206 as written: rhs/solution swapped, returns du = 0."""
    return np.zeros_like(dR_seed)


def solve_linear_fwd_intended(A: sp.csr_matrix, dR_seed: np.ndarray) -> np.ndarray:
    """What the docstring at fea_dolfinx.py:193-195 intends: du = A^{-1} dR."""
    return spla.splu(A.tocsc()).solve(dR_seed)


def total_gradient(mesh: OMesh, f: np.ndarray, u: np.ndarray, u_d: np.ndarray,
                   bc_dofs: np.ndarray, alpha: float = ALPHA_POISSON,
                   consistent_bc: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """dJ/df through the reverse sweep of SURVEY.md section 3.3:
        lam = A^{-T} dJ/du ;  dJ/df = dJ/df|partial - dRdf^T lam.
    With ``consistent_bc=False`` (the reference) dRdf keeps its Dirichlet rows
    (state_model.py:132-146 assemble without BCs) so lam on the boundary leaks
    into the gradient; ``True`` zeroes those rows (the mathematically exact
    reduced gradient, used only for the finite-difference check)."""
    lin = linearize(mesh, bc_dofs)
    dJdu = functional_du(mesh, u, u_d)
    lam = solve_linear_bwd(lin.A, dJdu)
    lam_used = lam.copy()
    if consistent_bc:
        lam_used[bc_dofs] = 0.0
    grad = functional_df(mesh, f, alpha) - lin.dRdf.T @ lam_used
    return grad, lam


# --------------------------------------------------------------------------
# Jacobi-preconditioned CG exactly as the HIP solver runs it (new design, not
# reference behaviour: the reference uses MUMPS LU; SURVEY.md section 0 finding 3)
# --------------------------------------------------------------------------
def pcg_jacobi(A: sp.csr_matrix, b: np.ndarray, x0: Optional[np.ndarray] = None,
               rtol: float = 1e-12, atol: float = 0.0, max_it: int = 100000
               ) -> Tuple[np.ndarray, int, float]:
    """Jacobi-PCG; stops on the natural norm sqrt(r^T D^-1 r) <= max(rtol sqrt(b^T D^-1 b), atol)
    (PETSc KSP_NORM_NATURAL [ext]), the rule of the HIP engine."""
    dinv = 1.0 / A.diagonal()
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    tol = max(rtol * math.sqrt(float(b @ (dinv * b))), atol)
    z = dinv * r
    rz = float(r @ z)
    if not math.sqrt(rz) > tol:
        return x, 0, math.sqrt(rz)
    p = z.copy()
    it = 0
    while it < max_it:
        q = A @ p
        pq = float(p @ q)
        alpha = rz / pq if pq != 0.0 else 0.0
        x += alpha * p
        r -= alpha * q
        it += 1
        z = dinv * r
        rz_new = float(r @ z)
        if math.sqrt(rz_new) <= tol:
            rz = rz_new
            break
        beta = rz_new / rz
        rz = rz_new
        p = z + beta * p
    return x, it, math.sqrt(rz)


# --------------------------------------------------------------------------
# DST-exact solve of the *discrete* P1 Poisson problem on un-jittered grids.
# On these meshes the assembled P1 stiffness equals the 5-point stencil (2-D)
# resp. h x the 7-point stencil (3-D); its interior
# block is diagonalised by the type-I discrete sine transform.
# --------------------------------------------------------------------------
def dst_solve_interior(n: int, tdim: int, rhs_interior: np.ndarray) -> np.ndarray:
    """Solve K_II x = rhs for the structured grid; rhs shaped (n-1,)*tdim (z,y,x)."""
    from scipy.fft import dstn, idstn
    h = 1.0 / n
    k = np.arange(1, n)
    lam1 = 2.0 - 2.0 * np.cos(np.pi * k * h)           # 1-D second-difference eigenvalues
    if tdim == 2:
        lam = lam1[:, None] + lam1[None, :]
    else:
        lam = h * (lam1[:, None, None] + lam1[None, :, None] + lam1[None, None, :])
    rh = dstn(rhs_interior, type=1)
    return idstn(rh / lam, type=1)


def dst_solve(mesh: OMesh, b: np.ndarray) -> np.ndarray:
    """x with x=b on the boundary rows (identity) and K_II x_I = b_I (homogeneous
    lifting assumed: boundary values of b must be 0 for an exact match with A)."""
    n, d = mesh.n, mesh.tdim
    shape = (n + 1,) * d
    B = b.reshape(shape)
    inner = (slice(1, n),) * d
    X = np.zeros(shape)
    X[inner] = dst_solve_interior(n, d, B[inner])
    out = X.ravel().copy()
    bdofs = boundary_vertices_box(mesh.x)
    out[bdofs] = b[bdofs]
    return out


def load_vector(mesh: OMesh, f: np.ndarray) -> np.ndarray:
    vol, _ = cell_geometry(mesh)
    F = np.zeros(mesh.n_vert)
    np.add.at(F, mesh.conn.ravel(), np.repeat(f * vol / (mesh.tdim + 1), mesh.tdim + 1))
    return F


# --------------------------------------------------------------------------
# The whole cycle the benchmark times, reference-faithful
# direct-solver flavour.  Used by tests (small n) and by bench.py's cpu_baseline
# "port" leg (bounded n).
# --------------------------------------------------------------------------
def reference_cycle(mesh: OMesh, f: np.ndarray, u_d: np.ndarray, bc_dofs: np.ndarray,
                    bc_vals: np.ndarray, alpha: float = ALPHA_POISSON) -> Dict[str, np.ndarray]:
    u, info = newton_solve(mesh, f, np.zeros(mesh.n_vert), bc_dofs, bc_vals)
    J = functional(mesh, u, f, u_d, alpha)
    grad, lam = total_gradient(mesh, f, u, u_d, bc_dofs, alpha)
    return dict(u=u, J=np.array([J]), grad=grad, lam=lam)


# ==========================================================================
# examples/nonlinear_poisson_opt: -div grad u + u^3 = f with symmetric Nitsche
# boundary terms  (run_nonlinear_poisson_opt.py:82-142, 196-210)
#   interior   inner(grad u, grad v) dx + inner(u**3, v) dx - inner(f, v) dx      (:88-96)
#   nitsche_1  - inner(dot(grad u, n), v) ds                                       (:109)
#   nitsche_2  sgn * inner(u_exact - u, dot(grad v, n)) ds,  sgn = +1 (sym=True)   (:110-111)
#   penalty    beta / h_E * inner(u - u_exact, v) ds,  beta = 10 (:98-100, :113-115)
# u_exact is the UFL expression sin(2 pi x) sin(pi y) in the reference (:145); its
# quadrature there uses an estimated degree [ext], which cannot be reproduced
# without FFCx.  Here u_exact is its CG1 interpolant, so every
# integrand is polynomial and integrated exactly: u^3 v by the closed-form P1
# monomial integrals (= any degree-4 rule), facet terms by the P1 facet mass matrix.
# h_E = UFL CellDiameter = largest vertex distance of the cell [ext].
# ==========================================================================
ALPHA_NL = 6e-7      # run_nonlinear_poisson_opt.py:80 ALPHA_1
BETA_NITSCHE = 10.0  # :98 beta_value=1e1


def u_exact_nl(x: np.ndarray) -> np.ndarray:
    """run_nonlinear_poisson_opt.py:145 (2-D); the 3-D analogue multiplies by sin(pi z)."""
    v = np.sin(2 * np.pi * x[:, 0]) * np.sin(np.pi * x[:, 1])
    if x.shape[1] == 3:
        v = v * np.sin(np.pi * x[:, 2])
    return v


def boundary_facets(mesh: OMesh) -> np.ndarray:
    """Bit k of mask[c] set <=> the facet of cell c opposite local vertex k lies on the boundary
    (it belongs to exactly one cell)."""
    d1 = mesh.tdim + 1
    nc = mesh.n_cell
    keys = []
    for k in range(d1):
        fv = np.sort(np.delete(mesh.conn, k, axis=1), axis=1).astype(np.int64)
        key = fv[:, 0]
        for j in range(1, fv.shape[1]):
            key = key * (mesh.n_vert + 1) + fv[:, j]
        keys.append(key)
    allk = np.concatenate(keys)
    _, inv, cnt = np.unique(allk, return_inverse=True, return_counts=True)
    onb = (cnt[inv] == 1).reshape(d1, nc)
    mask = np.zeros(nc, np.uint8)
    for k in range(d1):
        mask |= (onb[k].astype(np.uint8) << k)
    return mask


def _p1_cubic_tables(d: int):
    """T3[a,b,c,e] = int phi_a phi_b phi_c phi_e / |T| on a d-simplex (monomial formula
    int prod phi^alpha = |T| d! prod(alpha!) / (|alpha| + d)!)."""
    d1 = d + 1
    T = np.zeros((d1,) * 4)
    from itertools import product
    for idx in product(range(d1), repeat=4):
        mult = np.bincount(idx, minlength=d1)
        T[idx] = math.factorial(d) * np.prod([math.factorial(int(m)) for m in mult]) / math.factorial(4 + d)
    return T


def _facet_pieces(mesh: OMesh, bmask: np.ndarray):
    """Per (cell, local facet k) on the boundary: facet measure, outward unit normal, h_E."""
    vol, g = cell_geometry(mesh)
    d = mesh.tdim
    X = mesh.x[mesh.conn]
    hE = np.zeros(mesh.n_cell)
    for a in range(d + 1):
        for b in range(a + 1, d + 1):
            hE = np.maximum(hE, np.linalg.norm(X[:, a] - X[:, b], axis=1))
    out = []
    for k in range(d + 1):
        cells = np.nonzero((bmask >> k) & 1)[0]
        if cells.size == 0:
            continue
        gk = g[cells, k, :]
        ng = np.linalg.norm(gk, axis=1)
        nrm = -gk / ng[:, None]                       # outward normal of the facet opposite vertex k
        meas = d * vol[cells] * ng                    # |F_k| = d |T| |grad phi_k|
        out.append((k, cells, meas, nrm, hE[cells], g[cells]))
    return out


def nl_residual(mesh: OMesh, u: np.ndarray, f: np.ndarray, u_ex: np.ndarray, bmask: np.ndarray,
                beta: float = BETA_NITSCHE, sgn: float = 1.0) -> np.ndarray:
    """sgn = +1: sym=True (penalty on); sgn = -1 with beta = 0: the unsymmetric variant of :98-117."""
    d = mesh.tdim
    d1 = d + 1
    R = residual(mesh, u, f)                                            # grad-grad and load
    vol, _ = cell_geometry(mesh)
    T3 = _p1_cubic_tables(d)
    ue = u[mesh.conn]
    cub = vol[:, None] * np.einsum("abce,nb,nc,ne->na", T3, ue, ue, ue)
    np.add.at(R, mesh.conn.ravel(), cub.ravel())
    for k, cells, meas, nrm, hE, g in _facet_pieces(mesh, bmask):
        conn = mesh.conn[cells]
        on = [a for a in range(d1) if a != k]                            # facet vertices (local)
        e = (u - u_ex)[conn]                                             # (nf, d1)
        gn = np.einsum("nad,nd->na", g, nrm)                             # grad phi_a . n
        dun = np.einsum("na,na->n", gn, u[conn])                         # grad u . n
        Re = np.zeros((len(cells), d1))
        mean_e = e[:, on].sum(axis=1) / d                                # int_F e / |F|
        for a in range(d1):
            Re[:, a] += sgn * gn[:, a] * (-mean_e) * meas                # nitsche_2: sgn (g_a.n) int_F (u_ex - u)
        s_on = e[:, on].sum(axis=1)
        for a in on:
            Re[:, a] += -dun * meas / d                                  # nitsche_1
            Re[:, a] += beta / hE * meas / (d * (d + 1)) * (e[:, a] + s_on)   # penalty (facet mass)
        np.add.at(R, conn.ravel(), Re.ravel())
    return R


def nl_jacobian(mesh: OMesh, u: np.ndarray, bmask: np.ndarray, beta: float = BETA_NITSCHE, sgn: float = 1.0) -> sp.csr_matrix:
    d = mesh.tdim
    d1 = d + 1
    vol, g = cell_geometry(mesh)
    T3 = _p1_cubic_tables(d)
    ue = u[mesh.conn]
    Ke = vol[:, None, None] * (np.einsum("cad,cbd->cab", g, g) + 3.0 * np.einsum("abce,nc,ne->nab", T3, ue, ue))
    for k, cells, meas, nrm, hE, gg in _facet_pieces(mesh, bmask):
        on = [a for a in range(d1) if a != k]
        gn = np.einsum("nad,nd->na", gg, nrm)
        Fe = np.zeros((len(cells), d1, d1))
        for a in on:
            Fe[:, a, :] += -(gn * (meas / d)[:, None])                   # nitsche_1: -(g_b.n) int_F phi_a
            Fe[:, :, a] += -sgn * (gn * (meas / d)[:, None])             # nitsche_2: -sgn (g_a.n) int_F phi_b
            for b in on:
                Fe[:, a, b] += beta / hE * meas / (d * (d + 1)) * (2.0 if a == b else 1.0)
        Ke[cells] += Fe
    r = np.repeat(mesh.conn[:, :, None], d1, axis=2)
    c = np.repeat(mesh.conn[:, None, :], d1, axis=1)
    return _scatter_matrix(mesh, Ke, mesh.n_vert, mesh.n_vert, r, c)


def nl_newton_solve(mesh: OMesh, f: np.ndarray, u0: np.ndarray, u_ex: np.ndarray, bmask: np.ndarray,
                    beta: float = BETA_NITSCHE, atol: float = 1e-13, rtol: float = 1e-13, max_it: int = 100,
                    sgn: float = 1.0) -> Tuple[np.ndarray, SolveInfo]:
    """utils_dolfinx.py:376-416 SNES newtonls, line search basic (full step), LU; no strong BCs."""
    u = u0.copy()
    info = SolveInfo()
    F = nl_residual(mesh, u, f, u_ex, bmask, beta, sgn)
    r0 = float(np.linalg.norm(F))
    info.residual_norms.append(r0)
    while info.newton_its < max_it:
        r = info.residual_norms[-1]
        if r < atol or (info.newton_its > 0 and r < rtol * r0):
            break
        J = nl_jacobian(mesh, u, bmask, beta, sgn)
        u -= spla.splu(J.tocsc()).solve(F)
        info.newton_its += 1
        F = nl_residual(mesh, u, f, u_ex, bmask, beta, sgn)
        info.residual_norms.append(float(np.linalg.norm(F)))
    return u, info


def nl_reference_cycle(mesh: OMesh, f: np.ndarray, u_ex: np.ndarray, bmask: np.ndarray,
                       alpha: float = ALPHA_NL, beta: float = BETA_NITSCHE, sgn: float = 1.0) -> Dict[str, np.ndarray]:
    """run_nonlinear_poisson_opt.py: SNES solve from u = 1 (CSDL's default state value), J, adjoint
    gradient.  No Dirichlet rows, so A = dR/du and the reduced gradient is exact."""
    u, info = nl_newton_solve(mesh, f, np.ones(mesh.n_vert), u_ex, bmask, beta, sgn=sgn)
    J = functional(mesh, u, f, u_ex, alpha)
    dJdu = functional_du(mesh, u, u_ex)
    A = nl_jacobian(mesh, u, bmask, beta, sgn)
    lam = spla.splu(A.T.tocsc()).solve(dJdu)
    grad = functional_df(mesh, f, alpha) - dRdf(mesh).T @ lam
    return dict(u=u, J=np.array([J]), grad=grad, lam=lam, newton_its=info.newton_its)


# --------------------------------------------------------------------------
# L2 projection onto CG1 (utils_dolfinx.py:549-583), field outputs (fea_dolfinx.py:148-161)
# --------------------------------------------------------------------------
def mass_matrix(mesh: OMesh) -> sp.csr_matrix:
    vol, _ = cell_geometry(mesh)
    d1 = mesh.tdim + 1
    Me = (vol / (d1 * (d1 + 1)))[:, None, None] * (np.ones((d1, d1)) + np.eye(d1))[None]
    r = np.repeat(mesh.conn[:, :, None], d1, axis=2)
    c = np.repeat(mesh.conn[:, None, :], d1, axis=1)
    return _scatter_matrix(mesh, Me, mesh.n_vert, mesh.n_vert, r, c)


def project_l2(mesh: OMesh, cell_values: Optional[np.ndarray] = None, nodal_values: Optional[np.ndarray] = None,
               lump_mass: bool = False) -> np.ndarray:
    M = mass_matrix(mesh)
    b = load_vector(mesh, cell_values) if cell_values is not None else M @ nodal_values
    if lump_mass:
        return b / (M @ np.ones(mesh.n_vert))
    return spla.splu(M.tocsc()).solve(b)


def grad_magnitude(mesh: OMesh, u: np.ndarray) -> np.ndarray:
    _, g = cell_geometry(mesh)
    gu = np.einsum("cbd,cb->cd", g, u[mesh.conn])
    return np.sqrt((gu ** 2).sum(axis=1))


# ==========================================================================
# examples/beam_thickness_opt: Euler-Bernoulli cantilever, cubic Hermite elements
# (run_thickness_opt_cantilever_beam.py:41-162).  State = (w_0, th_0, w_1, th_1, ...),
# input = DG0 thickness per element, EI = E * width * t^3 / 12 (:71-75).
#   residual   inner(div grad v, EI div grad u) dx - f v(L)          (:77-79), f = -1 (:115)
#   compliance f u(L) (:84-85);   volume  t * width * L dx (:81-82)
# The element matrix is the textbook Hermite beam matrix (exact for the cubic shape
# functions; FFCx integrates the same polynomial exactly [ext]).
# GOLDEN VECTOR held by the reference: the 50 optimal thicknesses of the OpenMDAO example
# (:252-261) -- pinned in tests/test_oracle.py by re-running the optimisation with these
# gradients.
# ==========================================================================
def beam_khat(h: float) -> np.ndarray:
    return np.array([[12.0, 6 * h, -12.0, 6 * h],
                     [6 * h, 4 * h * h, -6 * h, 2 * h * h],
                     [-12.0, -6 * h, 12.0, -6 * h],
                     [6 * h, 2 * h * h, -6 * h, 4 * h * h]]) / h ** 3


def beam_stiffness(nel: int, L: float, t: np.ndarray, E: float = 1.0, width: float = 0.1) -> sp.csr_matrix:
    h = L / nel
    Kh = beam_khat(h)
    EI = E * width * t ** 3 / 12.0
    dofs = 2 * np.arange(nel)[:, None] + np.arange(4)[None, :]
    Ke = EI[:, None, None] * Kh[None]
    r = np.repeat(dofs[:, :, None], 4, axis=2)
    c = np.repeat(dofs[:, None, :], 4, axis=1)
    K = sp.coo_matrix((Ke.ravel(), (r.ravel(), c.ravel())), shape=(2 * nel + 2, 2 * nel + 2)).tocsr()
    K.sum_duplicates()
    K.sort_indices()
    return K


def beam_load(nel: int, f: float = -1.0) -> np.ndarray:
    F = np.zeros(2 * nel + 2)
    F[2 * nel] = f                      # point load on the deflection DOF of the end node
    return F


def beam_residual(nel, L, u, t, E=1.0, width=0.1, f=-1.0) -> np.ndarray:
    return beam_stiffness(nel, L, t, E, width) @ u - beam_load(nel, f)


def beam_dRdt(nel, L, u, t, E=1.0, width=0.1) -> sp.csr_matrix:
    """dR/dt: column e has the 4 entries (E width t_e^2 / 4) * Khat u_e."""
    h = L / nel
    Kh = beam_khat(h)
    dofs = 2 * np.arange(nel)[:, None] + np.arange(4)[None, :]
    vals = (E * width * t ** 2 / 4.0)[:, None] * (u[dofs] @ Kh.T)
    A = sp.coo_matrix((vals.ravel(), (dofs.ravel(), np.repeat(np.arange(nel), 4))), shape=(2 * nel + 2, nel)).tocsr()
    A.sort_indices()
    return A


def beam_cycle(nel, L, t, E=1.0, width=0.1, f=-1.0, consistent_bc=True):
    """Solve, compliance f u(L), volume, and d(compliance)/dt by the adjoint sweep (clamped at x = 0)."""
    bc = np.array([0, 1])
    K = beam_stiffness(nel, L, t, E, width)
    A = eliminate_bc(K, bc)
    F = beam_load(nel, f)
    b = F.copy()
    b[bc] = 0.0
    u = spla.splu(A.tocsc()).solve(b)
    compliance = float(F @ u)
    volume = float(np.sum(t) * width * (L / nel))
    lam = spla.splu(A.T.tocsc()).solve(F)                  # dJ/du = F
    if consistent_bc:
        lam[bc] = 0.0
    grad_c = -(beam_dRdt(nel, L, u, t, E, width).T @ lam)
    grad_v = np.full(nel, width * L / nel)
    return dict(u=u, compliance=compliance, volume=volume, grad_compliance=grad_c, grad_volume=grad_v, lam=lam)


BEAM_THICK_REF = np.array([   # run_thickness_opt_cantilever_beam.py:252-261 (OpenMDAO reference optimum)
    0.14915754, 0.14764328, 0.14611321, 0.14456715, 0.14300421, 0.14142417, 0.13982611, 0.13820976, 0.13657406,
    0.13491866, 0.13324268, 0.13154528, 0.12982575, 0.12808305, 0.12631658, 0.12452477, 0.12270701, 0.12086183,
    0.11898809, 0.11708424, 0.11514904, 0.11318072, 0.11117762, 0.10913764, 0.10705891, 0.10493903, 0.10277539,
    0.10056526, 0.09830546, 0.09599246, 0.09362243, 0.09119084, 0.08869265, 0.08612198, 0.08347229, 0.08073573,
    0.07790323, 0.07496382, 0.07190453, 0.06870925, 0.0653583, 0.06182632, 0.05808044, 0.05407658, 0.04975295,
    0.0450185, 0.03972912, 0.03363155, 0.02620192, 0.01610863])
