"""TEST INFRASTRUCTURE (oracle): CPU restatement of the Reissner-Mindlin shell of BASELINE config 3
(SURVEY.md section 8(f) row 3).  Only tests/ may import it; the product (femo_amd/) never does.

**Parity unpinned.**  The reference takes the weak form from the package ``shell_analysis_fenicsx``
(`examples/test_shell_m3l/shell_pde.py:5,246-253`: ``ShellElement``, ``MaterialModel``, ``ElasticModel.elasticEnergy``,
``weakFormResidual``), which is not in the tree and not installable here.  What the tree fixes is the discretisation
(`shell_pde.py:222-232`: element type "CG2CG1", separate measures ``dx_inplane`` / ``dx_shear``), the state
w = (u_mid in CG2^3, theta in CG1^3) (`shell_module.py`, `shell_pde.py:228`), the CG1 thickness (`shell_pde.py:229`)
and one known answer: the Scordelis-Lo roof, v_tip = -0.3024 (`examples/ongoing/shape_opt/run_shape_opt_roof.py:
48-51,131-160,224`: E = 4.32e8, nu = 0, h = 0.25, f = (0, 0, -90) per unit area, quarter model with a diaphragm at
x = 25 and two symmetry planes).  This file restates the published linear shell model that package is built on
(J. Bleyer, "Numerical tours of computational mechanics with FEniCS", linear shell demo [ext]: local tangent frame
per facet, membrane + bending + transverse shear + drilling energies, rotations as a global 3-vector) with the CG2/CG1
pair and a separate, lower-degree rule for the shear energy, and is pinned by that known answer (-0.2992 on a
16 x 16 mesh, -0.3010 on 32 x 32, converging from below), by a Kirchhoff plate solution and by rigid-body / patch
properties (tests/test_oracle_shell.py).

Kinematics on a flat facet with orthonormal tangent frame (e1, e2) and normal e3 (any tangent frame gives the same
energy for the isotropic material), s_j = tangent coordinates:
    eps_ij   = sym(e_i . du/ds_j)                                  membrane strain
    beta     = e3 x theta,  kappa_ij = sym(e_i . dbeta/ds_j)       bending strain
    gamma_j  = e3 . du/ds_j - e_j . beta                           transverse shear
    omega    = (e1 . du/ds_2 - e2 . du/ds_1) / 2 + e3 . theta      drilling strain (zero for rigid rotations)
Energy  1/2 int [ h eps:C:eps + h^3/12 kappa:C:kappa + E h^3 omega^2 ] dx_inplane + 1/2 int (5/6) mu h |gamma|^2 dx_shear,
C = plane-stress elasticity.  Degrees of freedom: 3 per P2 node (vertices, then edge midpoints), then 3 per vertex.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# Dunavant degree-4 rule (6 points): barycentric coordinates and weights (sum 1)
_A1, _B1, _W1 = 0.445948490915965, 0.108103018168070, 0.223381589678011
_A2, _B2, _W2 = 0.091576213509771, 0.816847572980459, 0.109951743655322
QUAD_INPLANE = (np.array([[_B1, _A1, _A1], [_A1, _B1, _A1], [_A1, _A1, _B1],
                          [_B2, _A2, _A2], [_A2, _B2, _A2], [_A2, _A2, _B2]]),
                np.array([_W1, _W1, _W1, _W2, _W2, _W2]))
# dx_shear: the degree-2 rule (3 points).  It is the lowest rule that keeps the CG2/CG1 pair rank sufficient -- with
# the one-point rule the assembled roof problem is singular (tests/test_oracle_shell.py) -- and it is "reduced"
# against dx_inplane (degree 4, needed for the CG1 thickness under the quadratic membrane terms).
QUAD_SHEAR = (np.array([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]]), np.array([1 / 3, 1 / 3, 1 / 3]))
QUAD_ONE_POINT = (np.array([[1 / 3, 1 / 3, 1 / 3]]), np.array([1.0]))
LOCAL_EDGES = ((0, 1), (1, 2), (2, 0))
SHEAR_CORRECTION = 5.0 / 6.0


class ShellSpace:
    """P2^3 x P1^3 on a triangulated surface.  ``conn``: (n_cell, 3) vertex indices."""

    def __init__(self, x: np.ndarray, conn: np.ndarray):
        self.x = np.asarray(x, dtype=np.float64)
        self.conn = np.asarray(conn, dtype=np.int64)
        nv = self.x.shape[0]
        pairs = np.stack([np.sort(self.conn[:, list(e)], axis=1) for e in LOCAL_EDGES], axis=1)      # (nc, 3, 2)
        key = pairs[..., 0] * nv + pairs[..., 1]
        uniq, inv = np.unique(key.ravel(), return_inverse=True)
        self.n_vert, self.n_edge = nv, uniq.size
        self.edge_vertices = np.stack([uniq // nv, uniq % nv], axis=1)
        self.cell_edges = inv.reshape(-1, 3)
        self.n_unode = nv + self.n_edge
        self.n_dof = 3 * self.n_unode + 3 * nv
        # element dof map: 6 displacement nodes x 3, then 3 rotation nodes x 3
        unodes = np.concatenate([self.conn, nv + self.cell_edges], axis=1)                               # (nc, 6)
        udofs = (3 * unodes[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 18)
        tdofs = (3 * self.n_unode + 3 * self.conn[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 9)
        self.cell_dofs = np.concatenate([udofs, tdofs], axis=1)                                          # (nc, 27)
        self.unode_x = np.concatenate([self.x, 0.5 * (self.x[self.edge_vertices[:, 0]] + self.x[self.edge_vertices[:, 1]])])

    # dof helpers ----------------------------------------------------------------------------------
    def u_dof(self, node, comp):
        return 3 * np.asarray(node) + comp

    def theta_dof(self, vertex, comp):
        return 3 * self.n_unode + 3 * np.asarray(vertex) + comp

    def vertex_displacement(self, w: np.ndarray) -> np.ndarray:
        return w[: 3 * self.n_vert].reshape(-1, 3)

    # geometry -------------------------------------------------------------------------------------
    def frames(self):
        """Per cell: e1, e2, e3, area, grad_lambda (3 barycentric gradients in tangent coordinates)."""
        p0, p1, p2 = (self.x[self.conn[:, k]] for k in range(3))
        t1, t2 = p1 - p0, p2 - p0
        n = np.cross(t1, t2)
        dbl = np.linalg.norm(n, axis=1)
        e3 = n / dbl[:, None]
        e1 = t1 / np.linalg.norm(t1, axis=1)[:, None]
        e2 = np.cross(e3, e1)
        area = 0.5 * dbl
        # tangent coordinates of the vertices: p0 = (0,0), p1 = (a,0), p2 = (b,c)
        a = np.einsum("ci,ci->c", t1, e1)
        b = np.einsum("ci,ci->c", t2, e1)
        c = np.einsum("ci,ci->c", t2, e2)
        X = np.stack([np.zeros_like(a), a, b], axis=1)
        Y = np.stack([np.zeros_like(a), np.zeros_like(a), c], axis=1)
        det = 2.0 * area
        gl = np.empty((self.conn.shape[0], 3, 2))
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            gl[:, i, 0] = (Y[:, j] - Y[:, k]) / det
            gl[:, i, 1] = (X[:, k] - X[:, j]) / det
        return e1, e2, e3, area, gl


def _p2(lam: np.ndarray, gl: np.ndarray):
    """P2 shape functions (6,) and their tangent gradients (nc, 6, 2) at barycentric point lam."""
    N = np.empty(6)
    dN = np.zeros((6, 3))                                           # d N_a / d lambda_i
    for i in range(3):
        N[i] = lam[i] * (2 * lam[i] - 1)
        dN[i, i] = 4 * lam[i] - 1
    for k, (i, j) in enumerate(LOCAL_EDGES):
        N[3 + k] = 4 * lam[i] * lam[j]
        dN[3 + k, i] = 4 * lam[j]
        dN[3 + k, j] = 4 * lam[i]
    return N, np.einsum("ai,cij->caj", dN, gl)


def _strain_operators(e1, e2, e3, gN, gM, M):
    """B matrices (nc, rows, 27) at one point: membrane (3, Voigt with engineering shear), bending (3), shear (2),
    drilling (1).  gN: (nc, 6, 2) P2 gradients, gM: (nc, 3, 2) P1 gradients, M: (3,) P1 values."""
    nc = e1.shape[0]
    Bm, Bb = np.zeros((nc, 3, 27)), np.zeros((nc, 3, 27))
    Bs, Bd = np.zeros((nc, 2, 27)), np.zeros((nc, 1, 27))
    for a in range(6):
        cols = slice(3 * a, 3 * a + 3)
        g1, g2 = gN[:, a, 0][:, None], gN[:, a, 1][:, None]
        Bm[:, 0, cols] = e1 * g1
        Bm[:, 1, cols] = e2 * g2
        Bm[:, 2, cols] = e1 * g2 + e2 * g1
        Bs[:, 0, cols] = e3 * g1
        Bs[:, 1, cols] = e3 * g2
        Bd[:, 0, cols] = 0.5 * (e1 * g2 - e2 * g1)
    for b in range(3):
        cols = slice(18 + 3 * b, 18 + 3 * b + 3)
        g1, g2 = gM[:, b, 0][:, None], gM[:, b, 1][:, None]
        # kappa_1j = -e2 . dtheta/ds_j,  kappa_2j = e1 . dtheta/ds_j
        Bb[:, 0, cols] = -e2 * g1
        Bb[:, 1, cols] = e1 * g2
        Bb[:, 2, cols] = -e2 * g2 + e1 * g1
        Bs[:, 0, cols] = e2 * M[b]
        Bs[:, 1, cols] = -e1 * M[b]
        Bd[:, 0, cols] = e3 * M[b]
    return Bm, Bb, Bs, Bd


def plane_stress(E: float, nu: float) -> np.ndarray:
    return E / (1.0 - nu * nu) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])


def element_stiffness(V: ShellSpace, h_nodal: np.ndarray, E: float, nu: float, return_parts: bool = False,
                      quad_shear=None):
    """(n_cell, 27, 27) element matrices for the CG1 thickness ``h_nodal`` (n_vert,)."""
    quad_shear = QUAD_SHEAR if quad_shear is None else quad_shear
    e1, e2, e3, area, gl = V.frames()
    C = plane_stress(E, nu)
    mu = E / (2.0 * (1.0 + nu))
    hc = np.asarray(h_nodal, dtype=np.float64)[V.conn]              # (nc, 3)
    nc = V.conn.shape[0]
    Km, Kb, Ks, Kd = (np.zeros((nc, 27, 27)) for _ in range(4))
    for lam, wq in zip(*QUAD_INPLANE):
        _, gN = _p2(lam, gl)
        Bm, Bb, _, Bd = _strain_operators(e1, e2, e3, gN, gl, lam)
        h = hc @ lam
        w = wq * area
        Km += np.einsum("c,cia,ij,cjb->cab", w * h, Bm, C, Bm)
        Kb += np.einsum("c,cia,ij,cjb->cab", w * h ** 3 / 12.0, Bb, C, Bb)
        Kd += np.einsum("c,cia,cib->cab", w * E * h ** 3, Bd, Bd)
    for lam, wq in zip(*quad_shear):
        _, gN = _p2(lam, gl)
        _, _, Bs, _ = _strain_operators(e1, e2, e3, gN, gl, lam)
        h = hc @ lam
        Ks += np.einsum("c,cia,cib->cab", wq * area * SHEAR_CORRECTION * mu * h, Bs, Bs)
    if return_parts:
        return Km, Kb, Ks, Kd
    return Km + Kb + Ks + Kd


def assemble(V: ShellSpace, Ke: np.ndarray) -> sp.csr_matrix:
    rows = np.repeat(V.cell_dofs, 27, axis=1).ravel()
    cols = np.tile(V.cell_dofs, (1, 27)).ravel()
    return sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(V.n_dof, V.n_dof)).tocsr()


def load_vector(V: ShellSpace, f_nodal: np.ndarray) -> np.ndarray:
    """int f . v with f a CG1 vector field given at the vertices (n_vert, 3) (force per unit area)."""
    _, _, _, area, gl = V.frames()
    F = np.zeros(V.n_dof)
    fc = np.asarray(f_nodal, dtype=np.float64)[V.conn]              # (nc, 3, 3)
    for lam, wq in zip(*QUAD_INPLANE):
        N, _ = _p2(lam, gl)
        fq = np.einsum("b,cbi->ci", lam, fc)
        contrib = np.einsum("c,a,ci->cai", wq * area, N, fq).reshape(-1, 18)
        np.add.at(F, V.cell_dofs[:, :18].ravel(), contrib.ravel())
    return F


def solve(K: sp.csr_matrix, F: np.ndarray, fixed: Sequence[int], values: Optional[np.ndarray] = None) -> np.ndarray:
    """Static solve with strongly imposed dofs (the reference imposes them strongly in run_shape_opt_roof.py:131-160
    and by penalty in shell_pde.py:246-253; the limit of the penalty form is this elimination)."""
    n = K.shape[0]
    fixed = np.unique(np.asarray(fixed, dtype=np.int64))
    w = np.zeros(n)
    if values is not None:
        w[fixed] = values
    free = np.setdiff1d(np.arange(n), fixed)
    rhs = F[free] - K[free][:, fixed] @ w[fixed]
    w[free] = spla.spsolve(K[free][:, free].tocsc(), rhs)
    return w


def energy_parts(V: ShellSpace, w: np.ndarray, h_nodal, E, nu) -> Dict[str, float]:
    parts = element_stiffness(V, h_nodal, E, nu, return_parts=True)
    we = w[V.cell_dofs]
    return {k: 0.5 * float(np.einsum("ca,cab,cb->", we, P, we)) for k, P in zip(("membrane", "bending", "shear", "drilling"), parts)}


def von_mises_stress(V: ShellSpace, w: np.ndarray, h_nodal, E: float, nu: float, surface: float = 1.0) -> np.ndarray:
    """von Mises stress of the in-plane stress sigma(z) = C (eps + z kappa) at z = surface * h / 2 ('Top' = +1, 'Mid' = 0,
    'Bot' = -1; shell_pde.py:315-328, whose ShellStressRM is in the absent shell_analysis_fenicsx: the standard
    Reissner-Mindlin recovery, transverse shear vanishing at the faces), at the six in-plane quadrature points: (nc, 6)."""
    e1, e2, e3, area, gl = V.frames()
    C = plane_stress(E, nu)
    hc = np.asarray(h_nodal, dtype=np.float64)[V.conn]
    we = np.asarray(w, dtype=np.float64)[V.cell_dofs]
    out = np.empty((V.conn.shape[0], len(QUAD_INPLANE[1])))
    for q, lam in enumerate(QUAD_INPLANE[0]):
        _, gN = _p2(lam, gl)
        Bm, Bb, _, _ = _strain_operators(e1, e2, e3, gN, gl, lam)
        z = 0.5 * surface * (hc @ lam)
        sig = np.einsum("ij,cj->ci", C, np.einsum("cia,ca->ci", Bm, we) + z[:, None] * np.einsum("cia,ca->ci", Bb, we))
        out[:, q] = np.sqrt(sig[:, 0] ** 2 - sig[:, 0] * sig[:, 1] + sig[:, 1] ** 2 + 3.0 * sig[:, 2] ** 2)
    return out


def project_von_mises(V: ShellSpace, w: np.ndarray, h_nodal, E: float, nu: float, surface: float = 1.0,
                      lump_mass: bool = False) -> np.ndarray:
    """L2 projection of the von Mises stress onto CG1 (shell_pde.py:330-332 `projected_von_Mises_stress`; the field
    output of the shell drivers, shell_dynamic_pde.py:82-83,129): M x = b, b_i = int sigma_vm phi_i (degree-4 rule),
    M the P1 mass matrix of the surface, or its row sums with ``lump_mass``."""
    _, _, _, area, _ = V.frames()
    vm = von_mises_stress(V, w, h_nodal, E, nu, surface)                  # (nc, 6)
    lam, wq = np.asarray(QUAD_INPLANE[0]), np.asarray(QUAD_INPLANE[1])
    be = np.einsum("c,q,cq,qi->ci", area, wq, vm, lam)
    b = np.zeros(V.n_vert)
    np.add.at(b, V.conn.ravel(), be.ravel())
    Me = area[:, None, None] / 12.0 * (np.ones((3, 3)) + np.eye(3))[None]
    M = sp.coo_matrix((Me.ravel(), (np.repeat(V.conn, 3, axis=1).ravel(), np.tile(V.conn, (1, 3)).ravel())),
                      shape=(V.n_vert, V.n_vert)).tocsc()
    if lump_mass:
        return b / np.asarray(M.sum(axis=1)).ravel()
    return spla.spsolve(M, b)


def pnorm_stress(V: ShellSpace, w: np.ndarray, h_nodal, E: float, nu: float, m: float = 1e-6, rho: float = 100.0,
                 alpha: Optional[float] = None, surface: float = 1.0, grad: bool = False):
    """1 / alpha int (m sigma_vm)^rho dx (shell_pde.py:297-313: the aggregated stress constraint of the shell drivers;
    alpha defaults to the surface area).  grad: also dJ/dw (n_dof,) and dJ/dh (n_vert,)."""
    e1, e2, e3, area, gl = V.frames()
    C = plane_stress(E, nu)
    hc = np.asarray(h_nodal, dtype=np.float64)[V.conn]
    we = np.asarray(w, dtype=np.float64)[V.cell_dofs]
    if alpha is None:
        alpha = float(area.sum())
    J = 0.0
    gw, gh = np.zeros(V.n_dof), np.zeros(V.n_vert)
    for lam, wq in zip(*QUAD_INPLANE):
        _, gN = _p2(lam, gl)
        Bm, Bb, _, _ = _strain_operators(e1, e2, e3, gN, gl, lam)
        z = 0.5 * surface * (hc @ lam)
        kap = np.einsum("cia,ca->ci", Bb, we)
        sig = np.einsum("ij,cj->ci", C, np.einsum("cia,ca->ci", Bm, we) + z[:, None] * kap)
        vm = np.sqrt(sig[:, 0] ** 2 - sig[:, 0] * sig[:, 1] + sig[:, 1] ** 2 + 3.0 * sig[:, 2] ** 2)
        J += float(np.sum(wq * area * (m * vm) ** rho)) / alpha
        if grad:
            safe = np.where(vm > 0.0, vm, 1.0)
            dvm = np.stack([2 * sig[:, 0] - sig[:, 1], 2 * sig[:, 1] - sig[:, 0], 6 * sig[:, 2]], axis=1) / (2.0 * safe[:, None])
            fac = np.where(vm > 0.0, wq * area * rho * m * (m * safe) ** (rho - 1.0) / alpha, 0.0)      # dJ / d vm
            ds = fac[:, None] * np.einsum("ci,ij->cj", dvm, C)                                          # dJ / d (eps + z kappa)
            np.add.at(gw, V.cell_dofs.ravel(), (np.einsum("ci,cia->ca", ds, Bm) + z[:, None] * np.einsum("ci,cia->ca", ds, Bb)).ravel())
            np.add.at(gh, V.conn.ravel(), (0.5 * surface * np.einsum("ci,ci->c", ds, kap)[:, None] * lam[None, :]).ravel())
    return (J, gw, gh) if grad else J


# ----------------------------------------------------------------------------------------------
# exact partials, boundary penalty, inertia, regularisation (round 3)
# ----------------------------------------------------------------------------------------------
def dform_dh(V: ShellSpace, h_nodal: np.ndarray, E: float, nu: float, v: np.ndarray, w: np.ndarray) -> np.ndarray:
    """g_b = v^T (dK/dh_b) w for the CG1 thickness, differentiated exactly: the thickness enters the quadrature-point
    factors h (membrane, shear) and h^3 (bending, drilling) through h_q = sum_b lam_b h_b.  This is (dR/dh)^T of
    compute_jacvec_product 'rev' (state_model.py:190-200) for the residual K(h) w - F."""
    e1, e2, e3, area, gl = V.frames()
    C = plane_stress(E, nu)
    mu = E / (2.0 * (1.0 + nu))
    hc = np.asarray(h_nodal, dtype=np.float64)[V.conn]
    ve, we = v[V.cell_dofs], w[V.cell_dofs]
    g = np.zeros(V.n_vert)
    for lam, wq in zip(*QUAD_INPLANE):
        _, gN = _p2(lam, gl)
        Bm, Bb, _, Bd = _strain_operators(e1, e2, e3, gN, gl, lam)
        h = hc @ lam
        mem = np.einsum("cia,ca,ij,cjb,cb->c", Bm, ve, C, Bm, we)
        ben = np.einsum("cia,ca,ij,cjb,cb->c", Bb, ve, C, Bb, we)
        dri = E * np.einsum("cia,ca,cib,cb->c", Bd, ve, Bd, we)
        d = wq * area * (mem + h * h * (0.25 * ben + 3.0 * dri))
        np.add.at(g, V.conn.ravel(), (d[:, None] * lam[None, :]).ravel())
    for lam, wq in zip(*QUAD_SHEAR):
        _, gN = _p2(lam, gl)
        _, _, Bs, _ = _strain_operators(e1, e2, e3, gN, gl, lam)
        sh = SHEAR_CORRECTION * mu * np.einsum("cia,ca,cib,cb->c", Bs, ve, Bs, we)
        np.add.at(g, V.conn.ravel(), ((wq * area * sh)[:, None] * lam[None, :]).ravel())
    return g


def p2_mass_apply(V: ShellSpace, coeff_nodal: Optional[np.ndarray], a: np.ndarray, cells: Optional[np.ndarray] = None,
                  power: int = 1) -> np.ndarray:
    """y = M a on the displacement dofs, M_ab = int c(x)^power N_a N_b (c CG1, given at the vertices; None: 1), degree-4
    rule; ``cells``: integrate over that subset only (a `dx(tag)` measure).  y has the length of the state vector."""
    _, _, _, area, gl = V.frames()
    y = np.zeros(V.n_dof)
    ue = a[V.cell_dofs[:, :18]].reshape(-1, 6, 3)
    sel = np.ones(V.conn.shape[0], bool) if cells is None else np.isin(np.arange(V.conn.shape[0]), cells)
    cc = None if coeff_nodal is None else np.asarray(coeff_nodal, dtype=np.float64)[V.conn]
    for lam, wq in zip(*QUAD_INPLANE):
        N, _ = _p2(lam, gl)
        c = 1.0 if cc is None else (cc @ lam) ** power
        uq = np.einsum("a,cai->ci", N, ue)
        contrib = np.einsum("c,a,ci->cai", wq * area * c * sel, N, uq).reshape(-1, 18)
        np.add.at(y, V.cell_dofs[:, :18].ravel(), contrib.ravel())
    return y


def compliance(V: ShellSpace, w: np.ndarray, cells: Optional[np.ndarray] = None) -> float:
    """1/2 int_dxx u_mid . u_mid (shell_pde.py:284-285 without the regularisation term), P2 mass matrix, degree-4 rule;
    ``cells``: the cells of the `dxx` measure the reference passes (`shell_pde.py:66`: dx_2(10), a tagged subset)."""
    return 0.5 * float(w @ p2_mass_apply(V, None, w, cells))


def compliance_du(V: ShellSpace, w: np.ndarray, cells: Optional[np.ndarray] = None) -> np.ndarray:
    """d compliance / d w = M_P2 u (zero on the rotations)."""
    return p2_mass_apply(V, None, w, cells)


def cell_diameter(V: ShellSpace) -> np.ndarray:
    """UFL CellDiameter [ext]: the largest vertex distance of the cell."""
    p = V.x[V.conn]
    return np.max(np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i, j in LOCAL_EDGES], axis=1), axis=1)


def regularization(V: ShellSpace, h_nodal: np.ndarray, kind: Optional[str] = None, grad: bool = False):
    """`ShellPDE.regularization(h, type)` (shell_pde.py:262-282), alpha1 = 1e3, alpha2 = 1:
        'H1'   1/2 alpha1 int |grad h|^2          'L2'  1/2 alpha1 int h^2
        'L2H1' 1/2 alpha1 int h^2 + 1/2 alpha2 int h_mesh^2 |grad h|^2        None: 0
    for the CG1 thickness on the flat facets (tangential gradient).  grad: also d/dh (n_vert,)."""
    h = np.asarray(h_nodal, dtype=np.float64)
    _, _, _, area, gl = V.frames()
    hc = h[V.conn]
    a1, a2 = 1e3, 1.0
    val, g = 0.0, np.zeros(V.n_vert)
    if kind is None:
        return (0.0, g) if grad else 0.0
    if kind not in ("H1", "L2H1", "L2"):
        raise ValueError(f"unknown regularisation {kind!r}")
    if kind in ("L2", "L2H1"):
        Me = area[:, None, None] / 12.0 * (np.ones((3, 3)) + np.eye(3))[None]          # P1 mass matrix
        Mh = np.einsum("cab,cb->ca", Me, hc)
        val += 0.5 * a1 * float((hc * Mh).sum())
        np.add.at(g, V.conn.ravel(), (a1 * Mh).ravel())
    if kind in ("H1", "L2H1"):
        gh = np.einsum("cbj,cb->cj", gl, hc)                                             # grad h per cell (tangent coordinates)
        coef = a1 * area if kind == "H1" else a2 * cell_diameter(V) ** 2 * area
        val += 0.5 * float((coef * (gh * gh).sum(axis=1)).sum())
        np.add.at(g, V.conn.ravel(), (coef[:, None] * np.einsum("cbj,cj->cb", gl, gh)).ravel())
    return (val, g) if grad else val


def tagged_edges(V: ShellSpace, marker) -> Tuple[np.ndarray, np.ndarray]:
    """Edges all of whose vertices satisfy ``marker(x)`` (x: (3, n) like dolfinx's locate_entities [ext]): (exterior edge
    ids, interior edge ids) -- the facets of the `ds(tag)` / `dS(tag)` measures of
    run_aeroelasticity_static_wo_feedback.py:110-124."""
    hit = np.asarray(marker(V.x.T), dtype=bool)
    on = hit[V.edge_vertices[:, 0]] & hit[V.edge_vertices[:, 1]]
    count = np.bincount(V.cell_edges.ravel(), minlength=V.n_edge)
    return np.nonzero(on & (count == 1))[0], np.nonzero(on & (count == 2))[0]


def penalty_matrix(V: ShellSpace, ext_edges: np.ndarray, int_edges: np.ndarray, beta: float) -> sp.csr_matrix:
    """Boundary penalty of `weakFormResidual(..., penalty=True, dss, dSS, g)` (shell_pde.py:246-253; the form itself is in
    the absent shell_analysis_fenicsx): restated in the idiom of the tree's other penalty terms
    (run_poisson_opt.py:60, motor_pde.py:177-178)
        beta / h_E (w - g) . dw  on ds,      [beta / h_E]('+') + [beta / h_E]('-') on dS,
    all six fields of w = (u_mid, theta), h_E = CellDiameter of the adjacent cell(s).  Edge mass matrices: P2 (two
    vertices + midpoint) for u, P1 for theta; exact (the reference integrates with degree 4).  Returns K_pen; the
    residual contribution is K_pen (w - g)."""
    hE = cell_diameter(V)
    inv_h = np.zeros(V.n_edge)
    np.add.at(inv_h, V.cell_edges.ravel(), np.repeat(1.0 / hE, 3))          # exterior: one cell; interior: both sides
    edges = np.concatenate([np.asarray(ext_edges, dtype=np.int64), np.asarray(int_edges, dtype=np.int64)])
    if edges.size == 0:
        return sp.csr_matrix((V.n_dof, V.n_dof))
    v0, v1 = V.edge_vertices[edges, 0], V.edge_vertices[edges, 1]
    length = np.linalg.norm(V.x[v1] - V.x[v0], axis=1)
    coef = beta * inv_h[edges] * length
    M2 = np.array([[4.0, -1.0, 2.0], [-1.0, 4.0, 2.0], [2.0, 2.0, 16.0]]) / 30.0      # P2 on [0, 1]: (v0, v1, midpoint)
    M1 = np.array([[2.0, 1.0], [1.0, 2.0]]) / 6.0
    un = np.stack([v0, v1, V.n_vert + edges], axis=1)                                   # displacement nodes of the edge
    rows, cols, vals = [], [], []
    for k in range(3):
        r = 3 * un + k
        rows.append(np.repeat(r, 3, axis=1).ravel()); cols.append(np.tile(r, (1, 3)).ravel())
        vals.append((coef[:, None, None] * M2[None]).ravel())
        t = 3 * V.n_unode + 3 * np.stack([v0, v1], axis=1) + k
        rows.append(np.repeat(t, 2, axis=1).ravel()); cols.append(np.tile(t, (1, 2)).ravel())
        vals.append((coef[:, None, None] * M1[None]).ravel())
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(V.n_dof, V.n_dof)).tocsr()
    K.sum_duplicates()
    return K


def inertia_apply(V: ShellSpace, h_nodal: np.ndarray, rho: float, acc: np.ndarray) -> np.ndarray:
    """Inertial residual of `kinetic_residual(rho, h)` (shell_pde.py:255-256 -> ElasticModel.inertialResidual [ext, absent]):
    the Reissner-Mindlin kinetic terms  int rho h  uddot . du  +  int rho h^3/12  thetaddot . dtheta  (consistent mass,
    degree-4 rule, CG1 thickness) applied to the acceleration vector ``acc`` = (uddot, thetaddot) in state layout."""
    _, _, _, area, _ = V.frames()
    h = np.asarray(h_nodal, dtype=np.float64)
    y = rho * p2_mass_apply(V, h, acc)
    hc = h[V.conn]
    te = acc[V.cell_dofs[:, 18:]].reshape(-1, 3, 3)
    for lam, wq in zip(*QUAD_INPLANE):
        hq = hc @ lam
        tq = np.einsum("b,cbi->ci", lam, te)
        contrib = np.einsum("c,b,ci->cbi", wq * area * rho * hq ** 3 / 12.0, lam, tq).reshape(-1, 9)
        np.add.at(y, V.cell_dofs[:, 18:].ravel(), contrib.ravel())
    return y


def reference_cycle(n: int, E: float = 4.32e8, nu: float = 0.0, h0: float = 0.25) -> Dict:
    """The cycle of BASELINE config 3 the way the reference runs it, on the n x n roof (bench.py's CPU leg): Newton with
    always three iterations, each assembling K and factorising it afresh (utils_dolfinx.py:419-449), compliance and
    dJ/dw, one more factorisation for the adjoint (state_model.py:157-158 with linear_problem = True), the transposed
    solve and dJ/dh = -lam^T dK/dh w."""
    pts, conn = scordelis_lo_mesh(n, n)
    V = ShellSpace(pts, conn)
    L = 25.0
    ux, vx = V.unode_x, V.x
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    fixed = np.unique(np.concatenate([
        V.u_dof(on(ux[:, 0], L), 1), V.u_dof(on(ux[:, 0], L), 2), V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0),
        V.theta_dof(on(vx[:, 1], 0.0), 2), V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))
    free = np.setdiff1d(np.arange(V.n_dof), fixed)
    h = np.full(V.n_vert, h0)
    F = load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1)))
    w = np.zeros(V.n_dof)
    for _ in range(3):
        K = assemble(V, element_stiffness(V, h, E, nu))
        r = K @ w - F
        lu = spla.splu(K[free][:, free].tocsc())
        w[free] -= lu.solve(r[free])
    J = compliance(V, w)
    dJdw = compliance_du(V, w)
    K = assemble(V, element_stiffness(V, h, E, nu))
    lu = spla.splu(K[free][:, free].tocsc())
    lam = np.zeros(V.n_dof)
    lam[free] = lu.solve(dJdw[free])
    grad = -dform_dh(V, h, E, nu, lam, w)
    tip = int(np.argmin(np.abs(vx[:, 0]) + np.abs(vx[:, 1] - vx[:, 1].max())))
    return dict(n_dof=int(V.n_dof), w=w, J=J, grad=grad, tip=float(V.vertex_displacement(w)[tip, 2]))


def scordelis_lo_mesh(nx: int, nphi: int, R: float = 25.0, L: float = 25.0, phi_max: float = np.deg2rad(40.0)):
    """Quarter of the Scordelis-Lo roof as in run_shape_opt_roof.py: axis along x in [0, L], y = R sin(phi),
    z = R cos(phi), phi in [0, 40 deg]; each (x, phi) cell split into two triangles."""
    xs = np.linspace(0.0, L, nx + 1)
    ph = np.linspace(0.0, phi_max, nphi + 1)
    X, P = np.meshgrid(xs, ph, indexing="ij")
    pts = np.stack([X.ravel(), R * np.sin(P).ravel(), R * np.cos(P).ravel()], axis=1)
    idx = np.arange((nx + 1) * (nphi + 1)).reshape(nx + 1, nphi + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    conn = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    return pts, conn


def scordelis_lo(nx: int, nphi: int) -> Tuple[float, ShellSpace, np.ndarray]:
    """Vertical displacement at the midpoint of the free edge (reference value -0.3024,
    run_shape_opt_roof.py:224) with the boundary conditions of run_shape_opt_roof.py:131-160."""
    E, nu, h, fz, L = 4.32e8, 0.0, 0.25, -90.0, 25.0
    pts, conn = scordelis_lo_mesh(nx, nphi, L=L)
    V = ShellSpace(pts, conn)
    K = assemble(V, element_stiffness(V, np.full(V.n_vert, h), E, nu))
    F = load_vector(V, np.tile([0.0, 0.0, fz], (V.n_vert, 1)))
    ux, vx = V.unode_x, V.x
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    fixed = np.concatenate([
        V.u_dof(on(ux[:, 0], L), 1), V.u_dof(on(ux[:, 0], L), 2),                       # diaphragm at x = L
        V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0), V.theta_dof(on(vx[:, 1], 0.0), 2),   # crown symmetry
        V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2),   # midspan symmetry
    ])
    w = solve(K, F, fixed)
    tip = int(np.argmin(np.abs(vx[:, 0]) + np.abs(vx[:, 1] - vx[:, 1].max())))
    return float(V.vertex_displacement(w)[tip, 2]), V, w


def plate_mesh(n: int, a: float = 1.0):
    """Flat square plate [0, a]^2 in the plane z = 0, n x n cells split into two triangles each."""
    g = np.linspace(0.0, a, n + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
    p, q, r, t = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([p, q, r], axis=1), np.stack([p, r, t], axis=1)])


def simply_supported_plate(n: int, h: float = 0.01, E: float = 1.0e7, nu: float = 0.3, q: float = -1.0):
    """Centre deflection of a simply supported square plate under uniform pressure; Kirchhoff series solution
    w = 0.00406235 q a^4 / D, D = E h^3 / (12 (1 - nu^2)) (Timoshenko & Woinowsky-Krieger, table 8 [ext])."""
    pts, conn = plate_mesh(n)
    V = ShellSpace(pts, conn)
    K = assemble(V, element_stiffness(V, np.full(V.n_vert, h), E, nu))
    F = load_vector(V, np.tile([0.0, 0.0, q], (V.n_vert, 1)))
    ux = V.unode_x
    edge = np.nonzero(np.isclose(ux[:, 0], 0) | np.isclose(ux[:, 0], 1) | np.isclose(ux[:, 1], 0) | np.isclose(ux[:, 1], 1))[0]
    fixed = np.concatenate([V.u_dof(edge, 2), V.u_dof(np.arange(V.n_unode), 0), V.u_dof(np.arange(V.n_unode), 1),
                            V.theta_dof(np.arange(V.n_vert), 2)])      # membrane and drilling dofs are decoupled here
    w = solve(K, F, fixed)
    centre = int(np.argmin(np.abs(V.x[:, 0] - 0.5) + np.abs(V.x[:, 1] - 0.5)))
    D = E * h ** 3 / (12.0 * (1.0 - nu * nu))
    return float(V.vertex_displacement(w)[centre, 2]), 0.00406235 * q / D


# ---- the lattice preconditioner of the shell's CG solves, restated in SciPy (round 4) -------------------------------------
# The preconditioner is this repository's own design (the reference factorises with MUMPS, shell_pde.py:246-253 through
# femo/fea/utils_dolfinx.py:476-512), so there is nothing in /root/reference to pin it against.  What the tests check: the
# HIP kernels apply exactly the operator written down here (transfers, Galerkin node blocks, dense coarse operator), the
# operator is symmetric positive definite, and the iteration counts below are pinned (tests/test_oracle_shell.py).
#   M^-1 = B_pt^-1 + w sum_{l > c} P_l B_l^-1 P_l^T + P_c (P_c^T K P_c)^-1 P_c^T,   w = level_weight = 0.3
# B_pt: 3 x 3 point blocks of K; B_l: 6 x 6 node blocks of P_l^T K P_l; P_l = P_L T_{L-1} ... T_l (nested lattices of
# 2, 4, ..., m cells per axis over the bounding cube).  hermite = True: the nodal rotations of a lattice are the slopes of
# its displacement interpolation (u = sum_n [alpha_n U_n + Theta_n x sigma_n], cubic Hermite shapes per axis), both from the
# mesh to the finest lattice and between the lattices; hermite = False: trilinear everywhere (rounds 2-3).
def lattice_levels(V: ShellSpace, finest: Optional[int] = None):
    pts = np.concatenate([V.unode_x, V.x])
    lo = pts.min(axis=0)
    ext = float((pts.max(axis=0) - lo).max()) * (1.0 + 1e-9) + 1e-300
    ev = V.edge_vertices
    h_avg = float(np.linalg.norm(V.x[ev[:, 0]] - V.x[ev[:, 1]], axis=1).mean())
    if finest is None:
        lg = np.log2(max(ext / h_avg, 2.0))
        finest = max(2, 2 ** int(np.floor(lg - 1.0 + 1e-9)), min(2 ** int(round(lg)), 32))
    levels, m = [], 2
    while m <= finest:
        levels.append(m)
        m *= 2
    return levels, lo, ext


def _lat_shapes(fr, H, hermite):
    lin = [1.0 - fr, fr]
    if not hermite:
        return lin, lin, [0.0 * fr, 0.0 * fr]
    return lin, [1.0 - 3 * fr ** 2 + 2 * fr ** 3, 3 * fr ** 2 - 2 * fr ** 3], [fr * (1.0 - fr) ** 2 * H, -(1.0 - fr) * fr ** 2 * H]


_EPS3 = (((0, 1, 2), 1.0), ((1, 2, 0), 1.0), ((2, 0, 1), 1.0), ((0, 2, 1), -1.0), ((2, 1, 0), -1.0), ((1, 0, 2), -1.0))


def lattice_node_sets(V: ShellSpace, levels, lo, ext):
    out = []
    for m in levels:
        ids = []
        for pts in (V.unode_x, V.x):
            t = (pts - lo) / ext * m
            i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
            for c in range(8):
                b = [(c >> k) & 1 for k in range(3)]
                ids.append(((i0[:, 2] + b[2]) * (m + 1) + i0[:, 1] + b[1]) * (m + 1) + i0[:, 0] + b[0])
        out.append(np.unique(np.concatenate(ids)))
    return out


def lattice_mesh_prolongation(V: ShellSpace, m: int, nodes: np.ndarray, lo, ext, hermite: bool) -> sp.csr_matrix:
    """P: unknowns (6 per lattice node of the compacted set ``nodes``) -> dofs, from the coordinates."""
    nu, nv, nd = V.n_unode, V.n_vert, V.n_dof
    rows, cols, vals = [], [], []
    H = ext / m

    def corner(points):
        t = (points - lo) / ext * m
        i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
        return i0, t - i0

    i0u, fu = corner(V.unode_x)
    i0t, ft = corner(V.x)
    su = [_lat_shapes(fu[:, k], H, hermite) for k in range(3)]
    st = [_lat_shapes(ft[:, k], H, False) for k in range(3)]
    pu, pt = np.arange(nu), np.arange(nv)
    for c in range(8):
        b = [(c >> k) & 1 for k in range(3)]
        gu = np.searchsorted(nodes, ((i0u[:, 2] + b[2]) * (m + 1) + i0u[:, 1] + b[1]) * (m + 1) + i0u[:, 0] + b[0])
        gt = np.searchsorted(nodes, ((i0t[:, 2] + b[2]) * (m + 1) + i0t[:, 1] + b[1]) * (m + 1) + i0t[:, 0] + b[0])
        w0 = su[0][1][b[0]] * su[1][1][b[1]] * su[2][1][b[2]]
        wt = st[0][0][b[0]] * st[1][0][b[1]] * st[2][0][b[2]]
        for i in range(3):
            rows.append(3 * pu + i); cols.append(6 * gu + i); vals.append(w0)
            rows.append(3 * nu + 3 * pt + i); cols.append(6 * gt + 3 + i); vals.append(wt)
        if hermite:
            for j in range(3):
                w1 = su[j][2][b[j]]
                for k in range(3):
                    if k != j:
                        w1 = w1 * su[k][1][b[k]]
                for (i, kk, jj), sgn in _EPS3:                                  # u_i += (Theta x e_j)_i w1 = eps_{i kk j} Theta_kk w1
                    if jj == j:
                        rows.append(3 * pu + i); cols.append(6 * gu + 3 + kk); vals.append(sgn * w1)
    P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, 6 * nodes.size))
    P.sum_duplicates()
    return P


def lattice_transfer(levels, sets, l: int, ext, hermite: bool) -> sp.csr_matrix:
    """T_l: unknowns of level l -> unknowns of level l + 1 (the coarse fields evaluated at the finer lattice's nodes)."""
    mc, mf = levels[l], levels[l + 1]
    g = sets[l + 1]
    ijk = np.stack([g % (mf + 1), (g // (mf + 1)) % (mf + 1), g // ((mf + 1) ** 2)], axis=1)
    t = ijk / 2.0
    i0 = np.clip(np.floor(t).astype(np.int64), 0, mc - 1)
    fr = t - i0
    sh = [_lat_shapes(fr[:, k], ext / mc, hermite) for k in range(3)]
    rows, cols, vals = [], [], []
    child = np.arange(g.size)
    for c in range(8):
        b = [(c >> k) & 1 for k in range(3)]
        gid = ((i0[:, 2] + b[2]) * (mc + 1) + i0[:, 1] + b[1]) * (mc + 1) + i0[:, 0] + b[0]
        w0 = sh[0][1][b[0]] * sh[1][1][b[1]] * sh[2][1][b[2]]
        wl = sh[0][0][b[0]] * sh[1][0][b[1]] * sh[2][0][b[2]]
        keep = wl > 0.0
        pos = np.searchsorted(sets[l], gid[keep])
        assert np.array_equal(sets[l][pos], gid[keep]), "lattice levels are not nested"
        kk = np.nonzero(keep)[0]
        for i in range(3):
            rows.append(6 * child[kk] + i); cols.append(6 * pos + i); vals.append(w0[kk])
            rows.append(6 * child[kk] + 3 + i); cols.append(6 * pos + 3 + i); vals.append(wl[kk])
        if hermite:
            for j in range(3):
                w1 = sh[j][2][b[j]]
                for k in range(3):
                    if k != j:
                        w1 = w1 * sh[k][1][b[k]]
                for (i, k2, jj), sgn in _EPS3:
                    if jj == j:
                        rows.append(6 * child[kk] + i); cols.append(6 * pos + 3 + k2); vals.append(sgn * w1[kk])
    T = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * g.size, 6 * sets[l].size))
    T.sum_duplicates()
    return T


def _block_diag_pinv(A: sp.csr_matrix, bs: int) -> sp.csr_matrix:
    nb = A.shape[0] // bs
    blocks = np.zeros((nb, bs, bs))
    coo = A.tocoo()
    sel = (coo.row // bs) == (coo.col // bs)
    np.add.at(blocks, (coo.row[sel] // bs, coo.row[sel] % bs, coo.col[sel] % bs), coo.data[sel])
    return blocks


class LatticePreconditioner:
    """z = M^-1 r as written above for the free-dof operator ``Kf`` (identity rows on the fixed dofs).  ``coarse_max``:
    unknowns of the exact-solve level (the finest level below the top with at most that many)."""

    def __init__(self, V: ShellSpace, K: sp.csr_matrix, fixed: Sequence[int], hermite: bool = True, finest: Optional[int] = None,
                 coarse_max: int = 3200, ridge: float = 1e-12, hermite_mesh: Optional[bool] = None, level_weight: float = 0.3):
        import scipy.sparse.linalg as spla
        nd = V.n_dof
        mask = np.ones(nd)
        mask[np.asarray(fixed, dtype=np.int64)] = 0.0
        Dm = sp.diags(mask)
        self.mask = mask
        # weight of the node-block levels in the additive sum (femo_shell_pc_weights: the overlapping levels overshoot with 1)
        self.level_weight = float(level_weight)
        self.Kf = (Dm @ K @ Dm + sp.diags(1.0 - mask)).tocsr()
        self.levels, lo, ext = lattice_levels(V, finest)
        self.sets = lattice_node_sets(V, self.levels, lo, ext)
        nl = len(self.levels)
        self.T = [lattice_transfer(self.levels, self.sets, l, ext, hermite) for l in range(nl - 1)]
        P = [None] * nl
        # hermite_mesh: the mesh <-> finest-lattice transfer on its own (experiments; default: like the lattice transfers)
        P[-1] = (Dm @ lattice_mesh_prolongation(V, self.levels[-1], self.sets[-1], lo, ext, hermite if hermite_mesh is None else hermite_mesh)).tocsr()
        for l in range(nl - 2, -1, -1):
            P[l] = (P[l + 1] @ self.T[l]).tocsr()
        self.P = P
        c = -1
        for l in range(nl - 1):
            if 6 * self.sets[l].size <= coarse_max:
                c = l
        self.c = c
        # point blocks (3 x 3) of the free-dof operator
        pb = _block_diag_pinv(self.Kf, 3)
        self.Bpt = np.linalg.inv(pb)
        # node blocks of the levels above c: symmetric 6 x 6, rows / columns no free dof reaches dropped
        self.Bl = {}
        for l in range(max(c, -1) + 1, nl):
            A = (P[l].T @ self.Kf @ P[l]).tocsr()
            blk = _block_diag_pinv(A, 6)
            inv = np.zeros_like(blk)
            for i in range(blk.shape[0]):
                d = np.diag(blk[i]) > 0.0
                if d.any():
                    sub = blk[i][np.ix_(d, d)] * (1.0 + 0.0) + np.diag(np.diag(blk[i])[d]) * ridge
                    inv[i][np.ix_(d, d)] = np.linalg.inv(sub)
            self.Bl[l] = inv
        self.Ac = None
        if c >= 0:
            Ac = (P[c].T @ self.Kf @ P[c]).toarray()
            d = np.diag(Ac).copy()
            # relative ridge as on the device (k_pc_coarse_fix_diag): the Hermite-type operator is semi-definite up to rounding
            # where the surface barely touches a node (its six unknowns are nearly dependent)
            Ac[np.diag_indices_from(Ac)] = np.where(d > 0.0, d * (1.0 + (1e-9 if hermite else 1e-13)), 1.0)
            self.Ac = Ac
            self.Ac_inv = np.linalg.inv(Ac)

    def apply(self, r: np.ndarray) -> np.ndarray:
        r = r * self.mask
        z = np.einsum("pij,pj->pi", self.Bpt, r.reshape(-1, 3)).ravel() * self.mask
        if self.c >= 0:
            z += self.P[self.c] @ (self.Ac_inv @ (self.P[self.c].T @ r))
        for l, B in self.Bl.items():
            g = (self.P[l].T @ r).reshape(-1, 6)
            z += self.level_weight * (self.P[l] @ np.einsum("nij,nj->ni", B, g).ravel())
        return z

    def pcg(self, b: np.ndarray, rtol: float = 1e-10, max_it: int = 5000):
        """PCG on the free-dof operator with the engine's stopping rule sqrt(r.z) <= rtol sqrt(r0.z0); returns (x, iterations)."""
        b = b * self.mask
        x = np.zeros_like(b)
        r = b.copy()
        z = self.apply(r)
        p = z.copy()
        g0 = g = float(r @ z)
        for it in range(1, max_it + 1):
            q = self.Kf @ p
            a = g / float(p @ q)
            x += a * p
            r -= a * q
            z = self.apply(r)
            g1 = float(r @ z)
            if not g1 > 0.0 or g1 <= rtol * rtol * g0:
                return x, it
            p = z + (g1 / g) * p
            g = g1
        return x, max_it
