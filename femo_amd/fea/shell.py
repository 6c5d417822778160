"""Reissner-Mindlin shell on the GPU engine: host side of `csrc/shell.hip`.

Mirrors the surface of `examples/test_shell_m3l/shell_pde.py:219-332` (``ShellPDE``: function spaces W / VT / VF,
``pdeRes``, ``compliance``, ``mass``, ``volume``, ``elastic_energy``) for the CG2^3 x CG1^3 element; the weak form the
reference imports from ``shell_analysis_fenicsx`` is restated in `oracle/shell_oracle.py` and implemented by the
kernels.  What is here: numbering of the P2 nodes, the CSR pattern of the 27 x 27 element couplings with the position
of every element entry (the arrays `femo_shell_create` takes), and ``ShellProblem``: residual, stiffness, forward and
adjoint solves, outputs and their partials -- the calls a StateOperation / OutputOperation makes
(`state_model.py:75-218`, `output_model.py:69-87`), with NumPy arrays at the boundary.

State layout (n_dof = 3 (n_vert + n_edge) + 3 n_vert): displacement of P2 node a, component k at 3 a + k (vertices first,
then edge midpoints), rotation of vertex v at 3 (n_vert + n_edge) + 3 v + k.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from .. import _lib
from .. import engine as E
from ..engine import Context, Vec, check

LOCAL_EDGES = ((0, 1), (1, 2), (2, 0))


class ShellSpace:
    """P2^3 x P1^3 degrees of freedom of a triangulated surface and the sparsity pattern of their couplings."""

    def __init__(self, x: np.ndarray, conn: np.ndarray):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        if self.x.ndim != 2 or self.x.shape[1] != 3 or self.conn.ndim != 2 or self.conn.shape[1] != 3:
            raise ValueError("ShellSpace: x must be (n_vert, 3) and conn (n_cell, 3)")
        nv = self.x.shape[0]
        c64 = self.conn.astype(np.int64)
        pairs = np.stack([np.sort(c64[:, list(e)], axis=1) for e in LOCAL_EDGES], axis=1)          # (nc, 3, 2)
        key = pairs[..., 0] * nv + pairs[..., 1]
        uniq, inv = np.unique(key.ravel(), return_inverse=True)
        self.n_vert, self.n_cell, self.n_edge = nv, self.conn.shape[0], int(uniq.size)
        self.edge_vertices = np.stack([uniq // nv, uniq % nv], axis=1)
        self.cell_edges = np.ascontiguousarray(inv.reshape(-1, 3), dtype=np.int32)
        self.n_unode = nv + self.n_edge
        self.n_dof = 3 * self.n_unode + 3 * nv
        unodes = np.concatenate([c64, nv + self.cell_edges.astype(np.int64)], axis=1)               # (nc, 6)
        udofs = (3 * unodes[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 18)
        tdofs = (3 * self.n_unode + 3 * c64[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 9)
        self.cell_dofs = np.concatenate([udofs, tdofs], axis=1)                                     # (nc, 27)
        self.unode_x = np.concatenate([self.x, 0.5 * (self.x[self.edge_vertices[:, 0]] + self.x[self.edge_vertices[:, 1]])])
        self._pattern = None

    def u_dof(self, node, comp):
        return 3 * np.asarray(node, dtype=np.int64) + comp

    def theta_dof(self, vertex, comp):
        return 3 * self.n_unode + 3 * np.asarray(vertex, dtype=np.int64) + comp

    def vertex_displacement(self, w: np.ndarray) -> np.ndarray:
        return np.asarray(w)[: 3 * self.n_vert].reshape(-1, 3)

    def pattern(self):
        """(rowptr int64, cols int32, elem_pos int32 (n_cell, 729)): CSR pattern of all element couplings (rows
        sorted by column) and the position of K_e[i][j] of every cell in it.  Built on the node blocks (a cell has 9:
        six P2 nodes and three rotation nodes, three dofs each): 81 instead of 729 sort keys per cell, the scalar
        pattern follows by arithmetic (`pattern_scalar_reference` is the direct construction the tests compare with)."""
        if self._pattern is None:
            nc, nbn = self.n_cell, self.n_dof // 3
            bn = self.cell_dofs[:, 0::3] // 3                                      # (nc, 9) block nodes of a cell
            key = (np.repeat(bn, 9, axis=1) * nbn + np.tile(bn, (1, 9))).ravel()
            uniq, inv = np.unique(key, return_inverse=True)
            if 9 * uniq.size >= 2 ** 31:
                raise ValueError("shell pattern exceeds 32-bit element positions")
            brow_of, bcol_of = uniq // nbn, uniq % nbn
            nb = np.bincount(brow_of, minlength=nbn)                               # blocks per block row
            brow = np.concatenate([[0], np.cumsum(nb)])
            slot = np.arange(uniq.size) - brow[brow_of]
            # scalar row 3 b + i starts at 9 brow[b] + 3 i nb[b]; block `slot` occupies 3 entries of it
            rowptr = np.empty(self.n_dof + 1, dtype=np.int64)
            rowptr[:-1] = (9 * brow[:-1][:, None] + 3 * nb[:, None] * np.arange(3)[None, :]).ravel()
            rowptr[-1] = 9 * uniq.size
            base = 9 * brow[brow_of] + 3 * slot                                    # entry (i = 0, j = 0) of every block
            stride = 3 * nb[brow_of]                                               # distance between its rows
            cols = np.empty(9 * uniq.size, dtype=np.int32)
            for i in range(3):
                for j in range(3):
                    cols[base + i * stride + j] = 3 * bcol_of + j
            inv = inv.reshape(nc, 9, 9)
            b32, s32 = base.astype(np.int32)[inv], stride.astype(np.int32)[inv]
            epos = np.empty((nc, 9, 3, 9, 3), dtype=np.int32)                       # (cell, ln, i, lm, j)
            for i in range(3):
                for j in range(3):
                    epos[:, :, i, :, j] = b32 + (i * s32 + j)
            self._pattern = (rowptr, cols, epos.reshape(nc, 729))
            self._block_lookup = (uniq, brow, nb, nbn)
        return self._pattern

    def positions(self, rows, cols) -> np.ndarray:
        """CSR positions of the couplings (rows[i], cols[i]) in the pattern (int32); every pair must be one."""
        self.pattern()
        uniq, brow, nb, nbn = self._block_lookup
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        br, bc = rows // 3, cols // 3
        b = np.searchsorted(uniq, br * nbn + bc)
        if np.any(b >= uniq.size) or np.any(uniq[np.minimum(b, uniq.size - 1)] != br * nbn + bc):
            raise ValueError("ShellSpace.positions: a pair is not in the element-coupling pattern")
        slot = b - brow[br]
        return (9 * brow[br] + 3 * (rows % 3) * nb[br] + 3 * slot + cols % 3).astype(np.int32)

    def cell_diameter(self) -> np.ndarray:
        """UFL CellDiameter [ext]: the largest vertex distance of each cell."""
        p = self.x[self.conn]
        return np.max(np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i, j in LOCAL_EDGES], axis=1), axis=1)

    def tagged_edges(self, marker):
        """(exterior edge ids, interior edge ids) of the edges all of whose vertices satisfy ``marker(x)`` (x: (3, n) as
        in dolfinx's locate_entities / locate_entities_boundary [ext]) -- the facets of the reference's `ds_1(100)` /
        `dS_1(100)` measures (run_aeroelasticity_static_wo_feedback.py:110-124)."""
        hit = np.asarray(marker(self.x.T), dtype=bool)
        on = hit[self.edge_vertices[:, 0]] & hit[self.edge_vertices[:, 1]]
        count = np.bincount(self.cell_edges.ravel(), minlength=self.n_edge)
        return np.nonzero(on & (count == 1))[0], np.nonzero(on & (count == 2))[0]

    def penalty_data(self, edges, beta: float):
        """What `femo_shell_set_penalty` takes for the tagged ``edges`` (ids, exterior and interior alike): per edge its
        three displacement nodes, coef = beta (sum over the adjacent cells of 1 / h_E) |edge| and the 39 CSR positions."""
        edges = np.asarray(edges, dtype=np.int64)
        inv_h = np.zeros(self.n_edge)
        np.add.at(inv_h, self.cell_edges.ravel(), np.repeat(1.0 / self.cell_diameter(), 3))
        v0, v1 = self.edge_vertices[edges, 0], self.edge_vertices[edges, 1]
        length = np.linalg.norm(self.x[v1] - self.x[v0], axis=1)
        coef = np.ascontiguousarray(beta * inv_h[edges] * length)
        un = np.stack([v0, v1, self.n_vert + edges], axis=1)                                  # (ne, 3)
        pos = np.empty((edges.size, 3, 13), dtype=np.int32)
        for k in range(3):
            r = 3 * un + k                                                                     # displacement dofs of the edge
            pos[:, k, :9] = self.positions(np.repeat(r, 3, axis=1).ravel(), np.tile(r, (1, 3)).ravel()).reshape(-1, 9)
            t = 3 * self.n_unode + 3 * un[:, :2] + k
            pos[:, k, 9:] = self.positions(np.repeat(t, 2, axis=1).ravel(), np.tile(t, (1, 2)).ravel()).reshape(-1, 4)
        return np.ascontiguousarray(un, dtype=np.int32), coef, np.ascontiguousarray(pos.reshape(-1, 39))

    def pattern_scalar_reference(self):
        """The same triple built directly from the 27 x 27 dof pairs of every cell (729 sort keys per cell)."""
        cd = self.cell_dofs
        rows = np.repeat(cd, 27, axis=1).ravel()
        cols = np.tile(cd, (1, 27)).ravel()
        uniq, inv = np.unique(rows * self.n_dof + cols, return_inverse=True)
        rowptr = np.zeros(self.n_dof + 1, dtype=np.int64)
        np.add.at(rowptr, uniq // self.n_dof + 1, 1)
        np.cumsum(rowptr, out=rowptr)
        return (rowptr, np.ascontiguousarray(uniq % self.n_dof, dtype=np.int32),
                np.ascontiguousarray(inv.reshape(-1, 729), dtype=np.int32))


def lattice_pc(space: ShellSpace, finest: Optional[int] = None):
    """Arrays of the lattice preconditioner (`femo_shell_pc_create`): nested lattices of 2, 4, ..., m cells per axis
    over the bounding cube, m = the largest power of two whose spacing is at least twice the mean edge length.  Per level the
    trilinear interpolation from the lattice nodes a dof node touches (compacted: a surface meets few nodes of a 3-D
    lattice), for each of the six fields (3 displacement components on the P2 nodes, 3 rotation components on the
    vertices).  Returns (width, n_lat, ell_idx (n_dof, width) int32, ell_w (n_dof, width), P^T as CSR)."""
    import scipy.sparse as sp
    pts = np.concatenate([space.unode_x, space.x])                    # displacement nodes, then rotation nodes
    n_pts = pts.shape[0]
    lo = pts.min(axis=0)
    ext = float((pts.max(axis=0) - lo).max()) * (1.0 + 1e-9) + 1e-300
    ev = space.edge_vertices
    h_avg = float(np.linalg.norm(space.x[ev[:, 0]] - space.x[ev[:, 1]], axis=1).mean())
    if finest is None:
        # about half the mesh density (P2 nodes: a quarter).  Measured on the roof at 62 k .. 1.97 M dofs: iteration counts
        # are flat from ext/h down to ext/(4 h) and double below, while every level dropped saves its launches and the
        # finest level's transfers get denser rows
        # -- but not below 32 cells per axis unless the mesh itself is coarser: on the 16 x 16 roof halving costs a third
        # more iterations (980 -> 1301)
        lg = np.log2(max(ext / h_avg, 2.0))
        finest = max(2, 2 ** int(np.floor(lg - 1.0 + 1e-9)), min(2 ** int(round(lg)), 32))
    levels = []
    m = 2
    while m <= finest:
        levels.append(m)
        m *= 2
    nl = len(levels)
    width = 8 * nl
    node_idx = np.zeros((n_pts, width), dtype=np.int64)
    node_w = np.zeros((n_pts, width))
    offsets, level_nodes = [], []
    off = 0
    for l, m in enumerate(levels):
        t = (pts - lo) / ext * m
        i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
        fr = t - i0
        gids = np.empty((n_pts, 8), dtype=np.int64)
        for c in range(8):
            w = np.ones(n_pts)
            ii = []
            for k in range(3):
                bit = (c >> k) & 1
                w = w * (fr[:, k] if bit else 1.0 - fr[:, k])
                ii.append(i0[:, k] + bit)
            gids[:, c] = (ii[2] * (m + 1) + ii[1]) * (m + 1) + ii[0]
            node_w[:, l * 8 + c] = w
        uniq, inv = np.unique(gids.ravel(), return_inverse=True)
        node_idx[:, l * 8:(l + 1) * 8] = off + inv.reshape(-1, 8)
        offsets.append(off)
        level_nodes.append(uniq)
        off += uniq.size
    offsets.append(off)
    n_nodes = off
    n_lat = 6 * n_nodes
    # dofs: field f of point p -> lattice unknown 6 * node + f
    nu, nv = space.n_unode, space.n_vert
    point_of = np.concatenate([np.repeat(np.arange(nu), 3), nu + np.repeat(np.arange(nv), 3)])
    field_of = np.concatenate([np.tile(np.arange(3), nu), 3 + np.tile(np.arange(3), nv)])
    ell_idx = (6 * node_idx[point_of] + field_of[:, None]).astype(np.int32)
    ell_w = np.ascontiguousarray(node_w[point_of])
    # finest level: P_L^T as CSR over all lattice unknowns (only the finest rows are non-empty)
    fin = slice(width - 8, width)
    rows = np.repeat(np.arange(space.n_dof), 8)
    P = sp.csr_matrix((ell_w[:, fin].ravel(), (rows, ell_idx[:, fin].ravel())), shape=(space.n_dof, n_lat))
    Pt = P.T.tocsr()
    Pt.sort_indices()
    # nested lattices: node-level transfer between consecutive levels (trilinear: even index -> one parent with
    # weight 1, odd -> two parents with weight 1/2, per axis).  Every parent of a touched node is itself touched
    # (it is a corner of the coarse cell that contains the point), so the compacted node sets are closed.
    pr, pc, pw = [], [], []
    for l in range(1, nl):
        m, mc = levels[l], levels[l - 1]
        g = level_nodes[l]                                        # global ids on the (m+1)^3 lattice, sorted
        i = g % (m + 1); j = (g // (m + 1)) % (m + 1); k = g // ((m + 1) ** 2)
        child = offsets[l] + np.arange(g.size)
        for c in range(8):
            w = np.ones(g.size)
            par = []
            for ax, q in enumerate((i, j, k)):
                bit = (c >> ax) & 1
                odd = (q & 1) == 1
                p_ax = np.where(odd, (q - 1) // 2 + bit, q // 2)       # odd: lower / upper parent, even: the one parent
                w = w * np.where(odd, 0.5, 1.0 if bit == 0 else 0.0)
                par.append(p_ax)
            keep = w > 0.0
            gp = (par[2] * (mc + 1) + par[1]) * (mc + 1) + par[0]
            pos = np.searchsorted(level_nodes[l - 1], gp[keep])
            if not np.array_equal(level_nodes[l - 1][pos], gp[keep]):
                raise RuntimeError("lattice levels are not nested")            # cannot happen, see above
            pr.append(child[keep]); pc.append(offsets[l - 1] + pos); pw.append(w[keep])
    if pr:
        Tp = sp.csr_matrix((np.concatenate(pw), (np.concatenate(pr), np.concatenate(pc))), shape=(n_nodes, n_nodes))
    else:
        Tp = sp.csr_matrix((n_nodes, n_nodes))
    Tp.sort_indices()
    Tc = Tp.T.tocsr()
    Tc.sort_indices()
    i64, i32 = np.int64, np.int32
    return dict(width=width, n_lat=int(n_lat), n_nodes=int(n_nodes), levels=levels, level_offsets=np.asarray(offsets, dtype=i64),
                level_nodes=level_nodes, n_unode=int(space.n_unode), lo=lo, ext=ext,
                ell_idx=np.ascontiguousarray(ell_idx), ell_w=ell_w,
                pt_rowptr=Pt.indptr.astype(i64), pt_cols=Pt.indices.astype(i32), pt_vals=np.ascontiguousarray(Pt.data),
                par_rowptr=Tp.indptr.astype(i64), par_cols=Tp.indices.astype(i32), par_vals=np.ascontiguousarray(Tp.data),
                chi_rowptr=Tc.indptr.astype(i64), chi_cols=Tc.indices.astype(i32), chi_vals=np.ascontiguousarray(Tc.data))


_CORNER_BITS = np.array([[(c >> k) & 1 for k in range(3)] for c in range(8)], dtype=np.int64)      # corner c of a cell: bit k = offset along axis k


def _hermite_1d(fr, H):
    """Per-axis shape values at fraction ``fr`` of a cell of size ``H``, for corner bit 0 / 1: the cubic Hermite value
    shapes h0, the slope shapes h1 (derivative 1 at their node, value 0 at both; the offset they multiply is signed) and
    the linear pair."""
    h0 = np.stack([1.0 - 3.0 * fr ** 2 + 2.0 * fr ** 3, 3.0 * fr ** 2 - 2.0 * fr ** 3], axis=-1)
    h1 = np.stack([fr * (1.0 - fr) ** 2 * H, -(1.0 - fr) * fr ** 2 * H], axis=-1)
    lin = np.stack([1.0 - fr, fr], axis=-1)
    return h0, h1, lin


def _hermite_corner_weights(h0, h1):
    """(alpha, sigma) of the eight corners from per-axis shapes h0, h1 of shape (n, 3, 2): alpha = prod_k h0_k,
    sigma_j = h1_j prod_{k != j} h0_k -- the interpolated displacement is u = sum_n [alpha_n U_n + Theta_n x sigma_n]."""
    n = h0.shape[0]
    alpha = np.empty((n, 8))
    sigma = np.empty((n, 8, 3))
    for c in range(8):
        b = _CORNER_BITS[c]
        a = [h0[:, k, b[k]] for k in range(3)]
        alpha[:, c] = a[0] * a[1] * a[2]
        sigma[:, c, 0] = h1[:, 0, b[0]] * a[1] * a[2]
        sigma[:, c, 1] = a[0] * h1[:, 1, b[1]] * a[2]
        sigma[:, c, 2] = a[0] * a[1] * h1[:, 2, b[2]]
    return alpha, sigma


def hermite_lattice(space: ShellSpace, L: dict, first_level: int = 0):
    """Hermite-type lattice spaces on the hierarchy of ``lattice_pc`` (round 4; DESIGN.md section 8 "the lead for round 4",
    prototype scripts/probe_shell_hermite.py): the nodal ROTATIONS of a lattice act as the slopes of its displacement
    interpolation,

        u(x) = sum_n [ alpha_n(x) U_n + Theta_n x sigma_n(x) ],   theta(x) = sum_n w_n(x) Theta_n  (trilinear),

    alpha = prod_k h0(t_k), sigma_j = h1(t_j) prod_{k != j} h0(t_k) with the cubic Hermite pair h0, h1 per axis, so that an
    inextensional bending mode (w quadratic, theta = grad w) is reproduced by coarse lattices whose spacing is far above
    the shell thickness -- where trilinear spaces lock.  Nested: only the finest lattice is interpolated from the mesh,
    level l + 1 from level l by the same shapes evaluated at the finer lattice's nodes (U_c = a U_p + Theta_p x b,
    Theta_c = c Theta_p).  The composition of such maps is again of the (alpha, sigma) form on the eight corners of the
    point's cell of the coarser level, so every level's COMPOSED prolongation costs four numbers per (displacement point,
    corner): what the Galerkin set-up kernels need (the level operators must be those of the transfers that are applied;
    the direct interpolation of a coarse level is NOT its composed one, scripts/probe_shell_hermite.py).

    Returns a dict: ``fin_w4`` (n_pts, 8, 4) float32 -- (alpha, sigma) of the displacement points on the finest lattice,
    (w, 0, 0, 0) for the rotation points; ``lvl_w4`` {level: (n_pts, 8, 4) float32} composed weights of the levels
    ``first_level`` .. finest - 1; ``par_w5`` / ``chi_w5`` (nnz, 5) -- (a, b, c) per entry of the node-level parent / child
    CSR of ``lattice_pc``; plus what `DeviceShell` derives from them."""
    import scipy.sparse as sp
    levels, off = L["levels"], L["level_offsets"]
    nl = len(levels)
    pts = np.concatenate([space.unode_x, space.x])
    n_pts, nu = pts.shape[0], space.n_unode
    lo, ext = L["lo"], L["ext"]

    def locate(m):
        t = (pts - lo) / ext * m
        i0 = np.clip(np.floor(t).astype(np.int64), 0, m - 1)
        return i0, t - i0

    # finest level: direct interpolation from the mesh
    mF = levels[-1]
    i0F, frF = locate(mF)
    h0, h1, lin = _hermite_1d(frF[:nu], ext / mF)
    alpha, sigma = _hermite_corner_weights(h0, h1)
    w_theta = L["ell_w"][3 * nu::3, 8 * (nl - 1):8 * nl]                 # rotation points: the trilinear weights as they are
    lvl = {nl - 1: (alpha, sigma)}
    i0_child = i0F[:nu]
    for l in range(nl - 2, first_level - 1, -1):
        # The transfer from level l to the corners of the point's cell of level l + 1 is a tensor product over the axes:
        # along axis k the fine cell is the lower or the upper half of the coarse one (half_k), its two nodes sit at
        # fractions (0, 1/2) or (1/2, 1) of the coarse cell, and a node's 1-D weights on the two coarse corners are the
        # shapes h0 / h1 / linear there.  alpha' = sum_child alpha prod_k A_k;  sigma'_j = sum_child [alpha B_j prod_{k != j} A_k
        # + sigma_j prod_k C_k]: separable contractions (einsum), 2 x 2 per axis.
        m, Hl = levels[l], ext / levels[l]
        i0 = locate(m)[0][:nu]
        half = i0_child - 2 * i0                                               # (nu, 3) in {0, 1}
        fr = np.stack([0.5 * half, 0.5 * half + 0.5], axis=-1)                 # (nu, 3, child bit): fraction inside the coarse cell
        h0, h1, lin = _hermite_1d(fr, Hl)                                      # (nu, 3, child bit, coarse corner bit)
        a_ch, s_ch = lvl[l + 1]
        A3 = a_ch.reshape(nu, 2, 2, 2)                                         # [bit of axis 2][axis 1][axis 0]
        S3 = s_ch.reshape(nu, 2, 2, 2, 3)

        def axis(T, M, ax):
            """Contract the child-bit axis ``ax`` (1, 2 or 3 of T[n, c2, c1, c0]: axis 3 is coordinate 0) with M[n, child bit, slot bit]."""
            idx0 = [slice(None)] * 4; idx1 = [slice(None)] * 4
            idx0[ax] = 0; idx1[ax] = 1
            t0, t1 = T[tuple(idx0)], T[tuple(idx1)]
            sh = (nu, 1, 1)
            return np.stack([t0 * M[:, 0, 0].reshape(sh) + t1 * M[:, 1, 0].reshape(sh),
                             t0 * M[:, 0, 1].reshape(sh) + t1 * M[:, 1, 1].reshape(sh)], axis=ax)

        Ak = [np.ascontiguousarray(h0[:, k]) for k in range(3)]
        Bk = [np.ascontiguousarray(h1[:, k]) for k in range(3)]
        Ck = [np.ascontiguousarray(lin[:, k]) for k in range(3)]
        # coordinate 0 first (array axis 3), shared partial products: A0, B0 -> A0A1, A0B1, B0A1 -> alpha and the three alpha-parts of sigma
        TA0, TB0 = axis(A3, Ak[0], 3), axis(A3, Bk[0], 3)
        TA0A1, TA0B1, TB0A1 = axis(TA0, Ak[1], 2), axis(TA0, Bk[1], 2), axis(TB0, Ak[1], 2)
        alpha = axis(TA0A1, Ak[2], 1).reshape(nu, 8)
        part = [axis(TB0A1, Ak[2], 1), axis(TA0B1, Ak[2], 1), axis(TA0A1, Bk[2], 1)]
        sigma = np.empty((nu, 8, 3))
        for j in range(3):
            Sj = np.ascontiguousarray(S3[..., j])
            sigma[:, :, j] = (part[j] + axis(axis(axis(Sj, Ck[0], 3), Ck[1], 2), Ck[2], 1)).reshape(nu, 8)
        lvl[l] = (alpha, sigma)
        i0_child = i0

    def pack(l):
        out = np.zeros((n_pts, 8, 4), dtype=np.float32)
        a, sg = lvl[l]
        out[:nu, :, 0] = a
        out[:nu, :, 1:] = sg
        out[nu:, :, 0] = L["ell_w"][3 * nu::3, 8 * l:8 * l + 8]
        return out

    fin_w4 = pack(nl - 1)
    lvl_w4 = {l: pack(l) for l in range(first_level, nl - 1)}
    # node-level transfers: (a, b, c) on the pattern of the trilinear parent CSR (same sparsity: h0(1/2) = 1/2, and the
    # slope shapes vanish wherever the trilinear weight does)
    nn = int(L["n_nodes"])
    comp = [[], [], [], [], []]
    pr, pc_ = [], []
    for l in range(1, nl):
        m, mc = levels[l], levels[l - 1]
        g = L["level_nodes"][l]
        ijk = np.stack([g % (m + 1), (g // (m + 1)) % (m + 1), g // ((m + 1) ** 2)], axis=1)
        t = ijk / 2.0
        p0 = np.clip(np.floor(t).astype(np.int64), 0, mc - 1)
        h0, h1, lin = _hermite_1d(t - p0, ext / mc)
        child = off[l] + np.arange(g.size)
        for P in range(8):
            b = _CORNER_BITS[P]
            a1 = [h0[:, k, b[k]] for k in range(3)]
            wl = lin[:, 0, b[0]] * lin[:, 1, b[1]] * lin[:, 2, b[2]]
            keep = wl > 0.0                                                    # the trilinear pattern
            par = p0 + b
            gp = (par[:, 2] * (mc + 1) + par[:, 1]) * (mc + 1) + par[:, 0]
            pos = np.searchsorted(L["level_nodes"][l - 1], gp[keep])
            pr.append(child[keep]); pc_.append(off[l - 1] + pos)
            comp[0].append((a1[0] * a1[1] * a1[2])[keep])
            comp[1].append((h1[:, 0, b[0]] * a1[1] * a1[2])[keep])
            comp[2].append((a1[0] * h1[:, 1, b[1]] * a1[2])[keep])
            comp[3].append((a1[0] * a1[1] * h1[:, 2, b[2]])[keep])
            comp[4].append(wl[keep])
    rows = np.concatenate(pr) if pr else np.zeros(0, np.int64)
    cols = np.concatenate(pc_) if pc_ else np.zeros(0, np.int64)
    # one CSR per component on identical index arrays (a marker matrix fixes the order scipy gives the entries)
    order = sp.csr_matrix((np.arange(1, rows.size + 1, dtype=np.float64), (rows, cols)), shape=(nn, nn))
    order.sort_indices()
    assert np.array_equal(order.indptr, L["par_rowptr"]) and np.array_equal(order.indices, L["par_cols"]), "Hermite transfers: pattern differs from the trilinear one"
    perm = order.data.astype(np.int64) - 1
    par_w5 = np.stack([np.concatenate(cp)[perm] for cp in comp], axis=1) if rows.size else np.zeros((0, 5))
    order_t = order.T.tocsr()
    order_t.sort_indices()
    assert np.array_equal(order_t.indptr, L["chi_rowptr"]) and np.array_equal(order_t.indices, L["chi_cols"])
    perm_t = order_t.data.astype(np.int64) - 1
    chi_w5 = np.stack([np.concatenate(cp)[perm_t] for cp in comp], axis=1) if rows.size else np.zeros((0, 5))
    return dict(fin_w4=fin_w4, lvl_w4=lvl_w4, par_w5=np.ascontiguousarray(par_w5), chi_w5=np.ascontiguousarray(chi_w5), w_theta=w_theta)


def hermite_transfer_matrix(L: dict, H: dict, l: int):
    """The node-level transfer T_l of ``hermite_lattice`` as a scalar sparse matrix over the 6 n unknowns (rows: unknowns of
    level l + 1, columns: of level l; global node numbering of ``lattice_pc``): U_c = a U_p + Theta_p x b, Theta_c = c Theta_p."""
    import scipy.sparse as sp
    nn = int(L["n_nodes"])
    rp, cols, w5 = L["par_rowptr"], L["par_cols"].astype(np.int64), H["par_w5"]
    rows = np.repeat(np.arange(nn), np.diff(rp))
    sel = (rows >= L["level_offsets"][l + 1]) & (rows < L["level_offsets"][l + 2])
    r, c, w = rows[sel], cols[sel], w5[sel]
    R, C, V = [], [], []
    for i in range(3):
        R.append(6 * r + i); C.append(6 * c + i); V.append(w[:, 0])
        R.append(6 * r + 3 + i); C.append(6 * c + 3 + i); V.append(w[:, 4])
    # (Theta x b)_0 = Th1 b2 - Th2 b1, (.)_1 = Th2 b0 - Th0 b2, (.)_2 = Th0 b1 - Th1 b0
    for (i, k, j, sg) in ((0, 1, 2, 1.0), (0, 2, 1, -1.0), (1, 2, 0, 1.0), (1, 0, 2, -1.0), (2, 0, 1, 1.0), (2, 1, 0, -1.0)):
        R.append(6 * r + i); C.append(6 * c + 3 + k); V.append(sg * w[:, 1 + j])
    return sp.csr_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(6 * nn, 6 * nn))


def hermite_device_arrays(space: ShellSpace, L: dict, cs_level: int) -> dict:
    """Everything `femo_shell_pc_hermite` takes, from ``hermite_lattice``: the finest level's P^T rows by lattice node, the
    composed weights of the levels above the coarse solve and of the coarse-solve level, the composite restriction from the
    finest lattice to the levels cs .. L - 2 in (A, B, C) form."""
    import scipy.sparse as sp
    H = hermite_lattice(space, L, first_level=cs_level)
    levels, off = L["levels"], L["level_offsets"]
    nl = len(levels)
    nu = space.n_unode
    n_pts = nu + space.n_vert
    fin_w4 = H["fin_w4"]
    # P_L^T: per finest node a row of displacement points and a row of rotation points
    node = (L["ell_idx"][0::3, 8 * (nl - 1):8 * nl].astype(np.int64) // 6) - off[nl - 1]          # (n_pts, 8) level-local node
    grp = (np.arange(n_pts) >= nu).astype(np.int64)
    key = (2 * node + grp[:, None]).ravel()
    pt = np.repeat(np.arange(n_pts), 8)
    w4 = fin_w4.reshape(-1, 4)
    keep = np.any(w4 != 0.0, axis=1)
    key, pt, w4 = key[keep], pt[keep], w4[keep]
    order = np.argsort(key, kind="stable")
    n_fin = int(off[nl] - off[nl - 1])
    hp_rowptr = np.zeros(2 * n_fin + 1, dtype=np.int64)
    np.add.at(hp_rowptr, key + 1, 1)
    hp_rowptr = np.cumsum(hp_rowptr)
    hp_cols = (3 * pt[order]).astype(np.int32)
    hp_w4 = np.ascontiguousarray(w4[order], dtype=np.float32)
    lvl_w4 = np.ascontiguousarray(np.stack([H["lvl_w4"][l] for l in range(cs_level + 1, nl - 1)] + [fin_w4]), dtype=np.float32)
    cs_w4 = np.ascontiguousarray(H["lvl_w4"][cs_level], dtype=np.float32)
    # composite restriction: R_l = (T_l ... T_{F-1})^T as 6 x 6 block rows, levels cs .. F - 1 stacked in node order
    F = nl - 1
    down = None
    if F > cs_level:
        nn = int(L["n_nodes"])
        comp, R = [], None
        for l in range(F - 1, cs_level - 1, -1):
            Tt = hermite_transfer_matrix(L, H, l).T.tocsr()                   # unknowns of level l + 1 -> level l (all node numbers global)
            rows = Tt[6 * off[l]:6 * off[l + 1]]
            R = rows if R is None else (rows[:, 6 * off[l + 1]:6 * off[l + 2]] @ R).tocsr()
            comp.append(R)                                                     # rows: level l's unknowns, columns: global (finest level's)
        stacked = sp.vstack(comp[::-1]).tocsr()
        B = sp.bsr_matrix(stacked, blocksize=(6, 6))
        B.sort_indices()
        d = B.data
        # block (parent rows, child columns): [UU] = A I, [TT] = C I, [TU] = [B]x = [[0, -B2, B1], [B2, 0, -B0], [-B1, B0, 0]]
        w5 = np.stack([d[:, 0, 0], d[:, 5, 1], d[:, 3, 2], d[:, 4, 0], d[:, 3, 3]], axis=1)
        down = dict(rowptr=B.indptr.astype(np.int64), cols=B.indices.astype(np.int32), w5=np.ascontiguousarray(w5))
    return dict(fin_w4=np.ascontiguousarray(fin_w4), hp_rowptr=hp_rowptr, hp_cols=hp_cols, hp_w4=hp_w4,
                par_w5=H["par_w5"], chi_w5=H["chi_w5"], lvl_w4=lvl_w4, cs_w4=cs_w4, down=down)


def node_block_items(L: dict, first_level: int, chunk: int = 64):
    """Arrays of `femo_shell_pc_block_items`: for the levels ``first_level`` .. finest of ``lattice_pc`` the points grouped by
    the lattice cell that contains them (all points of a cell share its eight nodes), cut into items of at most ``chunk``
    points, and the packed cell coordinates of every point per level."""
    levels, off = L["levels"], L["level_offsets"]
    n_pts = L["ell_idx"].shape[0] // 3
    ptrs, lvls, pts, cells = [np.zeros(1, dtype=np.int64)], [], [], []
    base = 0
    for j, l in enumerate(range(first_level, len(levels))):
        m, g = levels[l], L["level_nodes"][l]
        node0 = L["ell_idx"][0::3, 8 * l].astype(np.int64) // 6 - off[l]      # corner (0, 0, 0) of every point's cell
        gid = g[node0]
        if m + 1 > 1023:
            raise ValueError("lattice level too fine for the packed cell coordinates")
        cells.append((gid % (m + 1)) | (((gid // (m + 1)) % (m + 1)) << 10) | ((gid // ((m + 1) ** 2)) << 20))
        order = np.argsort(node0, kind="stable")
        skey = node0[order]
        starts = np.flatnonzero(np.r_[True, skey[1:] != skey[:-1]])
        ends = np.r_[starts[1:], skey.size]
        cuts = np.concatenate([np.arange(a, b, chunk) for a, b in zip(starts, ends)])
        ptrs.append(base + np.r_[cuts[1:], skey.size].astype(np.int64))
        lvls.append(np.full(cuts.size, j, dtype=np.int32))
        pts.append(order.astype(np.int32))
        base += n_pts
    return dict(item_ptr=np.ascontiguousarray(np.concatenate(ptrs)), item_lvl=np.ascontiguousarray(np.concatenate(lvls)),
                item_pts=np.ascontiguousarray(np.concatenate(pts)), pcell=np.ascontiguousarray(np.stack(cells).astype(np.int32)))


def coarse_solve_plan(L: dict, max_unknowns: int = 3200, chunk: int = 256):
    """Arrays of `femo_shell_pc_coarse` for the lattice levels of ``lattice_pc``: the coarse-solve level is the finest
    level (never the finest of the hierarchy) with at most ``max_unknowns`` unknowns; its points are grouped by (coarse
    cell, field group) and cut into items of at most ``chunk`` points.  None if no level qualifies."""
    import os
    chunk = int(os.environ.get("FEMO_SHELL_CG_CHUNK", chunk))
    levels, off = L["levels"], L["level_offsets"]
    c = -1
    for l in range(len(levels) - 1):
        if 6 * (off[l + 1] - off[l]) <= max_unknowns:
            c = l
    if c < 0:
        return None
    m, g = levels[c], L["level_nodes"][c]
    xyz = np.stack([g % (m + 1), (g // (m + 1)) % (m + 1), g // ((m + 1) ** 2)], axis=1).astype(np.int32)
    node0 = L["ell_idx"][0::3, 8 * c].astype(np.int64) // 6 - off[c]          # corner (0, 0, 0) of every point's cell
    n_pts = node0.size
    key = 2 * node0 + (np.arange(n_pts) >= L["n_unode"])
    order = np.argsort(key, kind="stable")
    skey = key[order]
    starts = np.flatnonzero(np.r_[True, skey[1:] != skey[:-1]])
    ends = np.r_[starts[1:], skey.size]
    cuts = np.concatenate([np.arange(a, b, chunk) for a, b in zip(starts, ends)])
    item_ptr = np.r_[cuts, skey.size].astype(np.int64)
    base = xyz[node0[order[cuts]]].astype(np.int64)                           # (n_items, 3)
    nbr = np.full((cuts.size, 64), -1, dtype=np.int32)
    for lz in range(4):
        for ly in range(4):
            for lx in range(4):
                q = base + np.array([lx - 1, ly - 1, lz - 1])
                ok = np.all((q >= 0) & (q <= m), axis=1)
                gid = (q[:, 2] * (m + 1) + q[:, 1]) * (m + 1) + q[:, 0]
                pos = np.minimum(np.searchsorted(g, gid), g.size - 1)
                ok &= g[pos] == gid
                nbr[ok, lz * 16 + ly * 4 + lx] = pos[ok]
    # composite restriction from the finest lattice to every level c .. F - 1 (g_l = T_l^T g_{l+1} chained on the host):
    # one launch per iteration instead of F - c
    import scipy.sparse as sp
    nn, F = int(L["n_nodes"]), len(levels) - 1
    Tc = sp.csr_matrix((L["chi_vals"], L["chi_cols"], L["chi_rowptr"]), shape=(nn, nn))
    comp, R = [], None
    for l in range(F - 1, c - 1, -1):
        rows = Tc[off[l]:off[l + 1]]                          # level l from level l + 1 (global column numbers)
        R = rows if R is None else (rows[:, off[l + 1]:off[l + 2]] @ R).tocsr()
        comp.append(R)
    down = sp.vstack(comp[::-1]).tocsr() if comp else None   # rows: levels c, c + 1, ..., F - 1 in node order
    if down is not None:
        down.sort_indices()
    return dict(level=int(c), node_xyz=np.ascontiguousarray(xyz), item_ptr=item_ptr,
                item_pts=np.ascontiguousarray(order, dtype=np.int32), item_nbr=np.ascontiguousarray(nbr),
                down_rowptr=None if down is None else down.indptr.astype(np.int64),
                down_cols=None if down is None else down.indices.astype(np.int32),
                down_vals=None if down is None else np.ascontiguousarray(down.data, dtype=np.float64))


class DeviceShell:
    """`femo_shell` handle."""

    def __init__(self, ctx: Context, space: ShellSpace, partition=None):
        """``partition``: a `dist.shell.ShellPartition` whose ``space`` is ``space`` -- the handle then holds one rank's part
        (`femo_shell_set_partition`); a context with several ranks takes nothing else."""
        if getattr(ctx, "nranks", 1) > 1 and partition is None:
            raise ValueError("a shell on a multi-rank context needs its partition (femo_amd.dist.shell.ShellPartition)")
        if partition is not None and partition.space is not space:
            raise ValueError("DeviceShell: the partition belongs to another space")
        self.ctx, self.space, self.lib, self.partition = ctx, space, _lib.load(), partition
        rowptr, cols, epos = space.pattern()
        self.handle = _lib.H()
        p = lambda a: C.c_void_p(a.ctypes.data)
        check(self.lib.femo_shell_create(ctx.handle, space.n_vert, p(space.x), space.n_cell, p(space.conn), space.n_edge,
                                         p(space.cell_edges), p(rowptr), p(cols), p(epos), C.byref(self.handle)))
        self.n_dof = int(self.lib.femo_shell_ndof(self.handle))
        self.nnz = int(self.lib.femo_shell_nnz(self.handle))
        assert self.n_dof == space.n_dof and self.nnz == cols.size
        self.pc_levels = None
        if partition is not None:
            P = partition
            q = lambda a: C.c_void_p(a.ctypes.data) if a.size else None
            check(self.lib.femo_shell_set_partition(self.handle, p(P.owned_points), int(P.nbr.size), q(P.nbr), p(P.send_ptr),
                                                    q(P.send_dofs), p(P.recv_ptr), q(P.recv_dofs)))

    def enable_lattice_pc(self, finest: Optional[int] = None, coarse_unknowns: Optional[int] = None, hermite: Optional[bool] = None) -> None:
        """Build and upload the lattice preconditioner once per mesh (used by ``solve(pc='lattice')``).
        ``coarse_unknowns``: size limit of the level that gets an exact (dense) coarse solve, 0 = none; default 3200
        (``FEMO_SHELL_COARSE`` overrides).  ``hermite`` (default on, one rank or partitioned, needs the coarse solve; ``FEMO_SHELL_TRILINEAR``
        switches it off): Hermite-type lattice spaces -- the rotations of a lattice as the slopes of its displacements."""
        import os
        if hermite is None:
            hermite = "FEMO_SHELL_TRILINEAR" not in os.environ
        if coarse_unknowns is None:
            coarse_unknowns = int(os.environ.get("FEMO_SHELL_COARSE", "3200"))
        if self.pc_levels is not None:
            return
        # partitioned: the global lattice (same nodes on every rank) with the rows of the local dofs
        L = lattice_pc(self.space, finest) if self.partition is None else self.partition.lattice(finest)
        p = lambda a: C.c_void_p(a.ctypes.data)
        check(self.lib.femo_shell_pc_create(self.handle, L["width"], L["n_nodes"], len(L["levels"]), p(L["level_offsets"]),
                                            p(L["ell_idx"]), p(L["ell_w"]), p(L["pt_rowptr"]), p(L["pt_cols"]), p(L["pt_vals"]),
                                            p(L["par_rowptr"]), p(L["par_cols"]), p(L["par_vals"]),
                                            p(L["chi_rowptr"]), p(L["chi_cols"]), p(L["chi_vals"])))
        self.pc_levels = L["levels"]
        self.coarse_level = None
        if coarse_unknowns > 0:
            plan = coarse_solve_plan(L, coarse_unknowns)
            if plan is not None:
                q = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
                check(self.lib.femo_shell_pc_coarse(self.handle, plan["level"], p(plan["node_xyz"]), plan["item_ptr"].size - 1,
                                                    p(plan["item_ptr"]), p(plan["item_pts"]), p(plan["item_nbr"]),
                                                    q(plan["down_rowptr"]), q(plan["down_cols"]), q(plan["down_vals"])))
                self.coarse_level = plan["level"]
        self.hermite = False
        if hermite and self.coarse_level is not None:
            A = hermite_device_arrays(self.space, L, self.coarse_level)
            q = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
            dn = A["down"] or {}
            check(self.lib.femo_shell_pc_hermite(self.handle, q(A["fin_w4"]), q(A["hp_rowptr"]), q(A["hp_cols"]), q(A["hp_w4"]), q(A["par_w5"]),
                                                 q(A["chi_w5"]), q(A["lvl_w4"]), q(A["cs_w4"]), q(dn.get("rowptr")), q(dn.get("cols")), q(dn.get("w5"))))
            self.hermite = True
            B = node_block_items(L, self.coarse_level + 1)
            check(self.lib.femo_shell_pc_block_items(self.handle, B["item_lvl"].size, q(B["item_ptr"]), q(B["item_lvl"]), q(B["item_pts"]), q(B["pcell"])))

    def pc_state(self) -> dict:
        """What the DEVICE side runs with (`femo_shell_pc_info`): ``self.hermite`` only says what was uploaded; the library
        falls back to the trilinear hierarchy when the Hermite-type coarse operator cannot be factorised."""
        out = (C.c_int32 * 4)()
        check(self.lib.femo_shell_pc_info(self.handle, out))
        return {"hermite_loaded": bool(out[0]), "hermite_enabled": bool(out[1] & 1), "hermite_in_use": bool(out[1] & 2),
                "fell_back_to_trilinear": bool(out[0]) and not bool(out[1] & 1),
                "coarse_solve_ready": bool(out[2]), "node_blocks_ready": bool(out[3])}

    LEVEL_WEIGHT = 0.3          # the library's default (femo_shell_pc_weights; oracle: LatticePreconditioner.level_weight)

    def pc_weights(self, level_weight: float = LEVEL_WEIGHT, coarse_weight: float = 1.0) -> None:
        """Weights of the node-block levels / of the exact coarse solve in the additive preconditioner (`femo_shell_pc_weights`);
        takes effect at the next solve."""
        check(self.lib.femo_shell_pc_weights(self.handle, float(level_weight), float(coarse_weight)))

    def pc_apply(self, vals: Vec, r: Vec, z: Vec, fixed: Optional[np.ndarray] = None) -> Vec:
        """z = M^-1 r of the lattice preconditioner for ``vals`` and the mask (`femo_shell_pc_apply`): for tests."""
        mask = np.ascontiguousarray(fixed, dtype=np.uint8) if fixed is not None else None
        check(self.lib.femo_shell_pc_apply(self.handle, vals.handle, C.c_void_p(mask.ctypes.data) if mask is not None else None, r.handle, z.handle))
        return z

    def coarse_matrix(self, vals: Vec, fixed: Optional[np.ndarray] = None, inverse: bool = False) -> np.ndarray:
        """The dense Galerkin operator of the coarse-solve level (or its inverse) for ``vals``: for tests."""
        n = C.c_int64(0)
        mask = np.ascontiguousarray(fixed, dtype=np.uint8) if fixed is not None else None
        mp = C.c_void_p(mask.ctypes.data) if mask is not None else None
        check(self.lib.femo_shell_pc_coarse_matrix(self.handle, vals.handle, mp, int(inverse), None, C.byref(n)))
        out = np.empty((n.value, n.value))
        check(self.lib.femo_shell_pc_coarse_matrix(self.handle, vals.handle, mp, int(inverse), C.c_void_p(out.ctypes.data), C.byref(n)))
        return out

    def __del__(self):
        # destroy only while the owning context is alive (interpreter shutdown tears objects down in arbitrary order)
        try:
            h, self.handle = getattr(self, "handle", None), None
            if h and getattr(self.ctx, "handle", None):
                self.lib.femo_shell_destroy(h)
        except Exception:
            pass

    # thin wrappers --------------------------------------------------------------------------------
    def assemble(self, Ey: float, nu: float, h: Vec, vals: Vec) -> Vec:
        check(self.lib.femo_shell_assemble(self.handle, float(Ey), float(nu), h.handle, vals.handle))
        return vals

    def set_owned_cells(self, owned) -> None:
        """Partitioned shells: the cells whose scalar outputs this rank integrates (`femo_shell_set_owned_cells`)."""
        a = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        check(self.lib.femo_shell_set_owned_cells(self.handle, C.c_void_p(a.ctypes.data) if a is not None else None))

    def halo(self, x: Vec) -> Vec:
        """x on the points owned by other ranks <- the owners' values (collective)."""
        check(self.lib.femo_shell_halo(self.handle, x.handle))
        return x

    def mask_unowned(self, x: Vec) -> Vec:
        """x <- 0 on the points owned by other ranks (no-op without a partition)."""
        check(self.lib.femo_shell_mask_unowned(self.handle, x.handle))
        return x

    def matvec(self, vals: Vec, x: Vec, y: Vec) -> Vec:
        check(self.lib.femo_shell_matvec(self.handle, vals.handle, None, x.handle, y.handle))
        return y

    def load(self, f: Vec, F: Vec, sign: float = 1.0, accumulate: bool = False) -> Vec:
        check(self.lib.femo_shell_load(self.handle, f.handle, float(sign), int(accumulate), F.handle))
        return F

    def load_T(self, lam: Vec, out: Vec, sign: float = 1.0, accumulate: bool = False) -> Vec:
        check(self.lib.femo_shell_load_T(self.handle, lam.handle, float(sign), int(accumulate), out.handle))
        return out

    def dform_dh(self, Ey, nu, h: Vec, v: Vec, w: Vec, out: Optional[Vec] = None, accumulate: bool = False, energy: bool = False):
        en = C.c_double(0.0)
        check(self.lib.femo_shell_dform_dh(self.handle, float(Ey), float(nu), h.handle, v.handle, w.handle, int(accumulate),
                                           out.handle if out is not None else None, C.byref(en) if energy else None))
        return en.value if energy else out

    def compliance(self, w: Vec, grad: Optional[Vec] = None, value: bool = True, accumulate: bool = False):
        val = C.c_double(0.0)
        check(self.lib.femo_shell_compliance(self.handle, w.handle, C.byref(val) if value else None, int(accumulate),
                                             grad.handle if grad is not None else None))
        return val.value

    def compliance_dx(self, w: Vec, cell_weight: Optional[Vec], grad: Optional[Vec] = None, value: bool = True, accumulate: bool = False):
        val = C.c_double(0.0)
        check(self.lib.femo_shell_compliance_dx(self.handle, w.handle, cell_weight.handle if cell_weight is not None else None,
                                                C.byref(val) if value else None, int(accumulate), grad.handle if grad is not None else None))
        return val.value

    REGULARIZATION_KINDS = {"H1": 1, "L2H1": 2, "L2": 3}

    def regularization(self, kind: str, h: Vec, grad: Optional[Vec] = None, value: bool = True, accumulate: bool = False) -> float:
        val = C.c_double(0.0)
        check(self.lib.femo_shell_regularization(self.handle, self.REGULARIZATION_KINDS[kind], h.handle, C.byref(val) if value else None,
                                                 int(accumulate), grad.handle if grad is not None else None))
        return val.value

    def hpower(self, coef: float, p: float, h: Vec, grad: Optional[Vec] = None, value: bool = True, accumulate: bool = False) -> float:
        val = C.c_double(0.0)
        check(self.lib.femo_shell_hpower(self.handle, float(coef), float(p), h.handle, C.byref(val) if value else None, int(accumulate),
                                         grad.handle if grad is not None else None))
        return val.value

    def set_penalty(self, edges, beta: float) -> None:
        """Tagged edges (ids) and penalty parameter of the boundary terms; ``edges`` empty: none."""
        edges = np.asarray(edges, dtype=np.int64)
        self._penalty_owner = None          # a direct call invalidates whatever form bound its edges last (shell_forms._bind_penalty)
        if edges.size == 0:
            check(self.lib.femo_shell_set_penalty(self.handle, 0, None, None, None))
            return
        un, coef, pos = self.space.penalty_data(edges, beta)
        p = lambda a: C.c_void_p(a.ctypes.data)
        check(self.lib.femo_shell_set_penalty(self.handle, edges.size, p(un), p(coef), p(pos)))

    def penalty_add(self, vals: Vec) -> Vec:
        check(self.lib.femo_shell_penalty_add(self.handle, vals.handle))
        return vals

    def penalty_apply(self, x: Vec, y: Vec, g: Optional[Vec] = None, accumulate: bool = False) -> Vec:
        check(self.lib.femo_shell_penalty_apply(self.handle, x.handle, g.handle if g is not None else None, int(accumulate), y.handle))
        return y

    def inertia_apply(self, rho: float, h: Vec, acc: Vec, y: Vec, accumulate: bool = False) -> Vec:
        check(self.lib.femo_shell_inertia_apply(self.handle, float(rho), h.handle, acc.handle, int(accumulate), y.handle))
        return y

    def inertia_dh(self, rho: float, h: Vec, lam: Vec, acc: Vec, out: Vec, accumulate: bool = False) -> Vec:
        check(self.lib.femo_shell_inertia_dh(self.handle, float(rho), h.handle, lam.handle, acc.handle, int(accumulate), out.handle))
        return out

    def inertia_dh_fwd(self, rho: float, h: Vec, dh: Vec, acc: Vec, y: Vec, accumulate: bool = False) -> Vec:
        """y (+)= (dM/dh [dh]) acc (`femo_shell_inertia_dh_fwd`)."""
        check(self.lib.femo_shell_inertia_dh_fwd(self.handle, float(rho), h.handle, dh.handle, acc.handle, int(accumulate), y.handle))
        return y

    def dform_dh_fwd(self, Ey, nu, h: Vec, dh: Vec, w: Vec, y: Vec, accumulate: bool = False) -> Vec:
        """y (+)= (dK/dh [dh]) w (`femo_shell_dform_dh_fwd`): the forward product with the thickness partial of the elastic residual."""
        check(self.lib.femo_shell_dform_dh_fwd(self.handle, float(Ey), float(nu), h.handle, dh.handle, w.handle, int(accumulate), y.handle))
        return y

    def mass(self, rho: float, h: Vec, grad: Optional[Vec] = None, value: bool = True, accumulate: bool = False):
        val = C.c_double(0.0)
        check(self.lib.femo_shell_mass(self.handle, float(rho), h.handle, C.byref(val) if value else None, int(accumulate),
                                       grad.handle if grad is not None else None))
        return val.value

    def pnorm_stress(self, Ey: float, nu: float, h: Vec, w: Vec, m: float, rho: float, alpha: float, surface: float = 1.0,
                     grad_w: Optional[Vec] = None, grad_h: Optional[Vec] = None, value: bool = True, accumulate: bool = False):
        val = C.c_double(0.0)
        check(self.lib.femo_shell_pnorm_stress(self.handle, float(Ey), float(nu), h.handle, w.handle, float(m), float(rho), float(alpha),
                                               float(surface), C.byref(val) if value else None, int(accumulate),
                                               grad_w.handle if grad_w is not None else None,
                                               grad_h.handle if grad_h is not None else None))
        return val.value

    def vm_rhs(self, Ey: float, nu: float, h: Vec, w: Vec, surface: float, rhs: Vec, lumped: Optional[Vec] = None) -> Vec:
        check(self.lib.femo_shell_vm_rhs(self.handle, float(Ey), float(nu), h.handle, w.handle, float(surface), rhs.handle,
                                         lumped.handle if lumped is not None else None))
        return rhs

    def p1_mass(self, x: Vec, y: Vec) -> Vec:
        check(self.lib.femo_shell_p1_mass(self.handle, x.handle, y.handle))
        return y

    def project_von_mises(self, Ey: float, nu: float, h: Vec, w: Vec, surface: float, out: Vec, lump_mass: bool = False,
                          rtol: float = 1e-12, max_it: int = 500) -> Vec:
        """L2 projection of the von Mises stress onto CG1 (`projected_von_Mises_stress`, shell_pde.py:330-332): lumped, or
        M x = b by CG preconditioned with the lumped mass (the P1 mass matrix: ~20 iterations whatever the mesh)."""
        nv, ctx = self.space.n_vert, self.ctx
        wk = self.__dict__.setdefault("_proj_work", None)
        if wk is None:
            wk = self._proj_work = [Vec(ctx, nv) for _ in range(6)]
        b, ml, r, z, p, q = wk
        self.vm_rhs(Ey, nu, h, w, surface, b, ml)
        part = self.partition
        if part is None:
            own = lambda v: v                       # one rank: every vertex is owned, nothing to refresh
            refresh = lambda v: v
            dot = lambda x, y: x.dot(y, nv)
        else:
            # Partitioned (round 5): a rank keeps every cell around the vertices it owns, so b, the lumped mass and the rows of the
            # P1 mass matrix are COMPLETE on owned vertices.  CG runs on the owned entries: residuals are masked (divided by a
            # vector that is 1 on owned vertices and inf on the others: x / inf = 0), dot products are all-reduced, and the
            # search direction is refreshed on the ghost vertices before every product -- it rides in the rotation dofs of a
            # state-sized vector through the shell's own halo plan (`femo_shell_halo`; vertex v = rotation point n_unode + v).
            # Host-staged: the projection is an output, not part of the solve loop.
            nu_l = self.space.n_unode
            if self.__dict__.get("_proj_own") is None:
                owned_v = np.asarray(part.owned_points[nu_l:nu_l + nv], dtype=bool)
                self._proj_own = Vec(ctx, nv).set(np.where(owned_v, 1.0, np.inf))
                self._proj_state = Vec(ctx, self.n_dof)
            own = lambda v: E.pointwise_divide(v, v, self._proj_own, nv)
            rot0 = 3 * nu_l + 3 * np.arange(nv)

            def refresh(v):
                st = np.zeros(self.n_dof)
                st[rot0] = np.asarray(v.get(nv))
                self._proj_state.set(st)
                self.halo(self._proj_state)
                v.set(np.ascontiguousarray(np.asarray(self._proj_state.get())[rot0]))
                return v
            dot = lambda x, y: float(ctx.allreduce_sum([x.dot(y, nv)])[0])       # x is masked: the rank's share
        if lump_mass:
            return refresh(E.pointwise_divide(out, b, ml, nv)) if part is not None else E.pointwise_divide(out, b, ml, nv)
        out.fill(0.0)
        r.copy_from(b)
        own(r)
        E.pointwise_divide(z, r, ml, nv)
        p.copy_from(z)
        g = g0 = dot(r, z)
        for _ in range(max_it):
            if g <= (rtol * rtol) * g0 or g0 == 0.0:
                return refresh(out) if part is not None else out
            refresh(p)
            self.p1_mass(p, q)
            own(q)
            own(p)
            a = g / dot(q, p)
            out.axpy(a, p)
            r.axpy(-a, q)
            E.pointwise_divide(z, r, ml, nv)
            g1 = dot(r, z)
            z.axpy(g1 / g, p)                    # z <- z + beta p, then p <- z
            p.copy_from(z)
            g = g1
        raise E.FemoError("the projection of the von Mises stress did not converge")

    def solve(self, vals: Vec, b: Vec, x: Vec, fixed: Optional[np.ndarray] = None, xfix: Optional[Vec] = None,
              rtol: float = 1e-12, atol: float = 0.0, max_it: int = 2_000_000, check_every: int = 64, pc: str = "lattice"):
        if pc == "lattice":
            self.enable_lattice_pc()
        elif pc != "jacobi":
            raise ValueError(f"unknown shell preconditioner {pc!r}")
        opts = _lib.SolverOpts(rtol=rtol, atol=atol, max_it=max_it, zero_guess=1, check_every=check_every,
                               pc=1 if pc == "lattice" else 0, atol_pc=0.0)
        info = _lib.SolveInfo()
        mask = None
        if fixed is not None:
            mask = np.ascontiguousarray(fixed, dtype=np.uint8)
            assert mask.size == self.n_dof
        check(self.lib.femo_shell_solve(self.handle, vals.handle, C.c_void_p(mask.ctypes.data) if mask is not None else None,
                                        xfix.handle if xfix is not None else None, b.handle, x.handle, C.byref(opts), C.byref(info)))
        if info.converged not in (1, 2):           # 2: stalled at the attainable accuracy, below 1e-9 relative (shell.hip)
            raise E.FemoError(f"shell CG did not converge: {info.iterations} iterations, residual {info.residual_norm:.3e} "
                              f"(rhs {info.rhs_norm:.3e})")
        return info


class ShellProblem:
    """The linear shell state problem R(w; h, f) = K(h) w - F(f) = 0 with strongly imposed dofs, its adjoint and the
    outputs of `shell_pde.py` -- what StateOperation / OutputOperation evaluate (`state_model.py:75-218`), NumPy at the
    boundary.  ``fixed_dofs``: indices into the state vector (the reference imposes them strongly in
    `run_shape_opt_roof.py:131-160` and by a penalty in `shell_pde.py:246-253`, whose limit this is)."""

    def __init__(self, x, conn, E_young: float, nu: float, fixed_dofs: Sequence[int] = (), ctx: Optional[Context] = None,
                 pc: str = "lattice", partition=None):
        """``partition`` (a `dist.shell.ShellPartition` of the space of ``x, conn``): this rank's part of the problem.  Inputs
        (thickness, load, fixed dofs) stay in GLOBAL numbering -- every rank passes the same arrays --, states and
        gradients come back on the rank's local points (`partition.dof_global / vert_global` map them; complete on the
        owned ones), scalar outputs are summed over the ranks."""
        from .utils_hip import get_context
        self.ctx = ctx if ctx is not None else get_context()
        self.partition = partition
        if partition is None:
            self.space = ShellSpace(x, conn)
        else:
            g = partition.global_space
            if g.x.shape != np.shape(x) or g.conn.shape != np.shape(conn):
                raise ValueError("ShellProblem: the partition was made for another mesh")
            self.space = partition.space
        self.dev = DeviceShell(self.ctx, self.space, partition)
        self.E, self.nu = float(E_young), float(nu)
        n, nv = self.space.n_dof, self.space.n_vert
        self.fixed = np.zeros(n, dtype=np.uint8)
        fd = np.asarray(list(fixed_dofs), dtype=np.int64)
        if partition is None:
            self.fixed[fd] = 1
        else:
            gfix = np.zeros(partition.global_space.n_dof, dtype=np.uint8)
            gfix[fd] = 1
            self.fixed[:] = gfix[partition.dof_global]
            self._cell_owned = Vec(self.ctx, self.space.n_cell).set(np.ascontiguousarray(partition.cell_owned, dtype=np.float64))
            self.dev.set_owned_cells(partition.cell_owned)              # scalar outputs: every cell of the whole mesh counted by one rank
        c = self.ctx
        self.h, self.f = Vec(c, nv), Vec(c, 3 * nv)
        self.w, self.F, self.tmp, self.lam = Vec(c, n), Vec(c, n), Vec(c, n), Vec(c, n)
        self.vals = Vec(c, self.dev.nnz)
        self.gh, self.gf = Vec(c, nv), Vec(c, 3 * nv)
        self._K_for = None
        self.last_info = None
        self.pc = pc

    # inputs ---------------------------------------------------------------------------------------
    def set_thickness(self, h) -> None:
        P = self.partition
        h = np.broadcast_to(np.asarray(h, dtype=np.float64), (self.space.n_vert if P is None else P.global_space.n_vert,))
        self.h.set(np.ascontiguousarray(h if P is None else h[P.vert_global]))
        self._K_for = None

    def set_load(self, f) -> None:
        P = self.partition
        f = np.broadcast_to(np.asarray(f, dtype=np.float64).reshape(-1, 3) if np.ndim(f) > 1 else np.asarray(f, dtype=np.float64),
                            (self.space.n_vert if P is None else P.global_space.n_vert, 3))
        self.f.set(np.ascontiguousarray(f if P is None else f[P.vert_global]).ravel())

    def _stiffness(self) -> Vec:
        if self._K_for is None:                     # reset by set_thickness
            self.dev.assemble(self.E, self.nu, self.h, self.vals)
            self._K_for = True
        return self.vals

    # state ----------------------------------------------------------------------------------------
    def residual(self, w: np.ndarray) -> np.ndarray:
        """K(h) w - F(f), no Dirichlet treatment (evaluate_residuals, state_model.py:75-85)."""
        K = self._stiffness()
        self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        self.dev.matvec(K, self.w, self.tmp)
        self.dev.load(self.f, self.tmp, sign=-1.0, accumulate=True)
        self.dev.mask_unowned(self.tmp)                        # partitioned: the rank's share (rows of its points)
        return E.writable(self.tmp.get())

    def solve(self, rtol: float = 1e-12) -> np.ndarray:
        """solve_residual_equations (state_model.py:87-115): w with the strongly imposed dofs at zero."""
        K = self._stiffness()
        self.dev.load(self.f, self.F)
        self.last_info = self.dev.solve(K, self.F, self.w, fixed=self.fixed, rtol=rtol, pc=self.pc)
        return E.writable(self.w.get())

    def solve_adjoint(self, rhs: np.ndarray, rtol: float = 1e-12) -> np.ndarray:
        """apply_inverse_jacobian 'rev' (state_model.py:202-218): K^-T rhs with the Dirichlet rows / columns eliminated
        (K is symmetric); the entries of rhs on imposed dofs do not enter."""
        K = self._stiffness()
        self.tmp.set(np.ascontiguousarray(rhs, dtype=np.float64))
        self.last_info = self.dev.solve(K, self.tmp, self.lam, fixed=self.fixed, rtol=rtol, pc=self.pc)
        return E.writable(self.lam.get())

    # partials of the residual --------------------------------------------------------------------
    def dRdh_T(self, lam: np.ndarray, w: Optional[np.ndarray] = None) -> np.ndarray:
        """(dR/dh)^T lam = lam^T dK/dh w per thickness dof (compute_jacvec_product 'rev', state_model.py:190-200)."""
        if w is not None:
            self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        self.lam.set(np.ascontiguousarray(lam, dtype=np.float64))
        self.dev.dform_dh(self.E, self.nu, self.h, self.lam, self.w, out=self.gh)
        return E.writable(self.gh.get())

    def dRdf_T(self, lam: np.ndarray) -> np.ndarray:
        """(dR/df)^T lam = -(dF/df)^T lam, shape (n_vert, 3)."""
        self.lam.set(np.ascontiguousarray(lam, dtype=np.float64))
        self.dev.load_T(self.lam, self.gf, sign=-1.0)
        return E.writable(self.gf.get()).reshape(-1, 3)

    # outputs --------------------------------------------------------------------------------------
    def compliance(self, w: Optional[np.ndarray] = None, grad: bool = False):
        if w is not None:
            self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        if self.partition is not None:
            # value: the cells this rank owns, summed over the ranks; gradient: all local cells (complete on owned points)
            J = self.dev.compliance_dx(self.w, self._cell_owned)
            if grad:
                self.dev.compliance(self.w, grad=self.tmp, value=False)
                return J, E.writable(self.tmp.get())
            return J
        if grad:
            J = self.dev.compliance(self.w, grad=self.tmp)
            return J, E.writable(self.tmp.get())
        return self.dev.compliance(self.w)

    def mass(self, rho: float = 1.0, grad: bool = False):
        """int rho h dx (shell_pde.py:287-294).  Partitioned: the value over the whole mesh (owned cells, summed over the ranks),
        the gradient on the rank's local vertices (complete on the owned ones)."""
        if grad:
            M = self.dev.mass(rho, self.h, grad=self.gh)
            return M, E.writable(self.gh.get())
        return self.dev.mass(rho, self.h)

    def surface_area(self) -> float:
        x, c = self.space.x, self.space.conn
        a = 0.5 * np.linalg.norm(np.cross(x[c[:, 1]] - x[c[:, 0]], x[c[:, 2]] - x[c[:, 0]]), axis=1)
        if self.partition is not None:                                   # the whole surface: owned cells, summed over the ranks
            return float(self.ctx.allreduce_sum([float(a[np.asarray(self.partition.cell_owned, dtype=bool)].sum())])[0])
        return float(a.sum())

    def pnorm_stress(self, w: Optional[np.ndarray] = None, m: float = 1e-6, rho: float = 100.0, alpha: Optional[float] = None,
                     surface: float = 1.0, grad: bool = False):
        """`ShellPDE.pnorm_stress` (shell_pde.py:297-313): 1 / alpha int (m sigma_vm)^rho dx on the top (surface = +1), mid (0) or
        bottom (-1) surface, alpha = surface area by default.  grad: also (dJ/dw, dJ/dh)."""
        if w is not None:
            self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        if alpha is None:
            alpha = self.surface_area()
        if grad:
            J = self.dev.pnorm_stress(self.E, self.nu, self.h, self.w, m, rho, alpha, surface, grad_w=self.tmp, grad_h=self.gh)
            return J, E.writable(self.tmp.get()), E.writable(self.gh.get())
        return self.dev.pnorm_stress(self.E, self.nu, self.h, self.w, m, rho, alpha, surface)

    def von_mises_field(self, w: Optional[np.ndarray] = None, surface: float = 1.0, lump_mass: bool = False) -> np.ndarray:
        """The von Mises stress on the top / mid / bottom surface projected onto the vertices (shell_pde.py:315-332)."""
        if w is not None:
            self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        self.dev.project_von_mises(self.E, self.nu, self.h, self.w, surface, self.gh, lump_mass=lump_mass)
        return E.writable(self.gh.get())

    def elastic_energy(self, w: Optional[np.ndarray] = None) -> float:
        if w is not None:
            self.w.set(np.ascontiguousarray(w, dtype=np.float64))
        return self.dev.dform_dh(self.E, self.nu, self.h, self.w, self.w, energy=True)

    # the adjoint cycle of BASELINE config 3 ----------------------------------------------------
    def compliance_gradient(self):
        """J(h) = 1/2 int |u_mid(h)|^2 and dJ/dh by the adjoint: K w = F, K lam = dJ/dw, dJ/dh = -lam^T dK/dh w."""
        w = self.solve()
        J, dJdw = self.compliance(grad=True)
        dJdw[self.fixed.astype(bool)] = 0.0
        lam = self.solve_adjoint(dJdw)
        return J, -self.dRdh_T(lam), w
