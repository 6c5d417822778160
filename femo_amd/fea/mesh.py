"""Host-side P1 simplex meshes (replaces dolfinx.mesh for the hot path).

``createUnitSquareMesh`` mirrors femo/fea/utils_dolfinx.py:136-140 (dolfinx
``create_unit_square``, right diagonals [ext]); ``createUnitCubeMesh`` is the 3-D
analogue the BASELINE.json configs need (dolfinx ``create_unit_cube``: six
tetrahedra per cell around the main diagonal [ext]).  Vertices are numbered
lexicographically (x fastest); DOF numbering == vertex numbering.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


class Mesh:
    """tdim in {2, 3}; ``x`` (n_vert, tdim) float64, ``conn`` (n_cell, tdim+1) int32."""

    def __init__(self, x: np.ndarray, conn: np.ndarray, n: int = 0):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.tdim = self.x.shape[1]
        if self.conn.shape[1] != self.tdim + 1:
            raise ValueError("connectivity width must be tdim+1 (P1 simplices)")
        self.n = n
        self._device = None
        self._ctx = None

    @property
    def n_vert(self) -> int:
        return self.x.shape[0]

    @property
    def n_cell(self) -> int:
        return self.conn.shape[0]

    def device(self, ctx):
        """The DeviceMesh (incidence + sparsity pattern) of this mesh on ``ctx``."""
        if self._device is None or self._ctx is not ctx:
            from ..engine import DeviceMesh
            self._device = DeviceMesh(ctx, self.x, self.conn)
            self._ctx = ctx
        return self._device

    def centroids(self) -> np.ndarray:
        return self.x[self.conn].mean(axis=1)

    def permuted(self, seed: int = 0, cells: bool = False) -> "Mesh":
        """The same mesh with a random vertex numbering (and, with ``cells``, a random cell order): what
        an unstructured mesh generator without bandwidth reduction hands over.  No SELL slice is regular,
        every gather is scattered.  ``vertex_perm[v]`` is the new index of old vertex v."""
        rng = np.random.default_rng(seed)
        p = rng.permutation(self.n_vert).astype(np.int32)
        x = np.empty_like(self.x)
        x[p] = self.x
        conn = p[self.conn]
        cp = rng.permutation(self.n_cell) if cells else None
        if cp is not None:
            conn = conn[cp]
        m = Mesh(x, conn, self.n)
        m.vertex_perm, m.cell_perm = p, cp
        return m

    def numbering_quality(self) -> float:
        """Fraction of the operator pattern's SELL slices that are regular in this numbering (columns = row + per-slice
        deltas, so the SpMV fetches no column indices; host-only topology build).  ~1 for structured numberings such as
        a lexicographic grid, 0 for a generator's or a random numbering."""
        from ..engine import topology_host
        info, _, _ = topology_host(self.tdim, self.n_vert, self.n_vert, self.conn)
        return info["regular_slices"] / max(info["n_slices"], 1)

    def reordered(self, force: bool = False) -> "Mesh":
        """Renumbered for locality unless the numbering it comes with is already a structured one: a mesh whose slices
        are mostly regular (``numbering_quality() >= 0.5``) streams its operators faster than any space-filling order
        could (round 3, 10 M-DOF cube: 153 M DOFs/s in lexicographic numbering, 138 M along the Morton curve, because the
        Morton order has no regular slice at all) and is returned unchanged, with identity maps; ``force=True``
        renumbers regardless.

        The same mesh renumbered for locality, as dolfinx does to every mesh it reads (it reorders the dofs and
        keeps `input_global_indices` / `original_cell_index` [ext]): vertices along a Morton (Z-order) curve through
        their coordinates, cells by their smallest vertex.  Rows of the operators then gather from a few cache lines
        instead of the whole vector: on the randomly numbered 10 M-DOF cube the SpMV fetches 18.9 GB per launch,
        1.6 GB after reordering.  ``original_vertex_index[i]`` / ``original_cell_index[c]`` give the input numbering;
        ``vertex_perm[v]`` is the new index of input vertex v."""
        d = self.tdim
        if not force and self.n_vert >= 64 and self.numbering_quality() >= 0.5:
            m = Mesh(self.x, self.conn, self.n)
            m.vertex_perm = np.arange(self.n_vert, dtype=np.int32)
            m.original_vertex_index = np.arange(self.n_vert, dtype=np.int64)
            m.original_cell_index = np.arange(self.n_cell, dtype=np.int64)
            return m
        lo, hi = self.x.min(axis=0), self.x.max(axis=0)
        bits = 21 if d == 3 else 31
        q = ((self.x - lo) / np.where(hi > lo, hi - lo, 1.0) * ((1 << bits) - 1)).astype(np.uint64)

        def spread(v):                      # insert d-1 zero bits between the bits of v
            if d == 3:
                v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
                v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
                v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
                v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
                v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
            else:
                v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
                v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
                v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
                v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
                v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
            return v

        code = np.zeros(self.n_vert, dtype=np.uint64)
        for k in range(d):
            code |= spread(q[:, k]) << np.uint64(k)
        order = np.argsort(code, kind="stable")                  # new vertex i = input vertex order[i]
        perm = np.empty(self.n_vert, dtype=np.int32)
        perm[order] = np.arange(self.n_vert, dtype=np.int32)
        conn = perm[self.conn]
        corder = np.argsort(conn.min(axis=1), kind="stable")
        m = Mesh(self.x[order], conn[corder], self.n)
        m.vertex_perm = perm
        m.original_vertex_index = order.astype(np.int64)
        m.original_cell_index = corder.astype(np.int64)
        return m

    def lattice_occupancy(self) -> float:
        """Largest over mean number of vertices in the non-empty bins of the finest BPX lattice: a
        measure of mesh grading.  The lattice hierarchy has no levels between its finest spacing
        (~2 average mesh sizes) and the local mesh size, so on strongly graded meshes BPX loses to
        Jacobi (measured: ratio 4-12 -> 2-6x fewer iterations than Jacobi, ratio 35-40 -> 1.3-1.7x
        more); the solver layer uses it only while this ratio stays below BPX_MAX_OCCUPANCY."""
        if getattr(self, "_occupancy", None) is None:
            from ..engine import pc_plan_host
            self._occupancy = pc_plan_host(self.x)["occupancy"]
        return self._occupancy

    def boundary_facet_mask(self) -> np.ndarray:
        """uint8 per cell: bit k set <=> the facet opposite local vertex k belongs to one cell only,
        i.e. it is an exterior facet (the `ds` measure of UFL [ext])."""
        if getattr(self, "_bfacets", None) is None:
            d1 = self.tdim + 1
            # all facets as sorted vertex tuples, (d1 * n_cell, tdim); equal tuples are found by a
            # lexicographic sort and a comparison of neighbours (no packing of the tuple into one
            # integer: (n_vert + 1) ** tdim overflows int64 beyond ~2 M vertices in 3-D)
            fv = np.concatenate([np.sort(np.delete(self.conn, k, axis=1), axis=1) for k in range(d1)], axis=0)
            order = np.lexsort(tuple(fv[:, j] for j in range(fv.shape[1] - 1, -1, -1)))
            sf = fv[order]
            same_next = np.zeros(len(sf), dtype=bool)
            same_next[:-1] = np.all(sf[1:] == sf[:-1], axis=1)
            shared = same_next.copy()
            shared[1:] |= same_next[:-1]
            once_flat = np.empty(len(sf), dtype=bool)
            once_flat[order] = ~shared
            once = once_flat.reshape(d1, self.n_cell)
            mask = np.zeros(self.n_cell, dtype=np.uint8)
            for k in range(d1):
                mask |= once[k].astype(np.uint8) << np.uint8(k)
            self._bfacets = mask
        return self._bfacets


def _apply_jitter(x: np.ndarray, n: int, jitter: float, seed: int) -> np.ndarray:
    """Seeded interior perturbation x += jitter*h*U(-1,1)."""
    if jitter == 0.0:
        return x
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1.0, 1.0, size=x.shape) * (jitter / n)
    interior = np.all((x > 1e-9) & (x < 1.0 - 1e-9), axis=1)
    x = x.copy()
    x[interior] += d[interior]
    return x


def createUnitSquareMesh(n: int, jitter: float = 0.0, seed: int = 20240807) -> Mesh:
    np1 = n + 1
    g = np.arange(np1) / n
    g[-1] = 1.0
    x = np.empty((np1 * np1, 2))
    x[:, 0] = np.tile(g, np1)
    x[:, 1] = np.repeat(g, np1)
    jj, ii = np.divmod(np.arange(n * n), n)
    v0 = jj * np1 + ii
    conn = np.empty((2 * n * n, 3), dtype=np.int32)
    conn[0::2, 0] = v0; conn[0::2, 1] = v0 + 1; conn[0::2, 2] = v0 + np1 + 1
    conn[1::2, 0] = v0; conn[1::2, 1] = v0 + np1 + 1; conn[1::2, 2] = v0 + np1
    return Mesh(_apply_jitter(x, n, jitter, seed), conn, n)


def createCylindricalRoofMesh(nx: int, nphi: int, R: float = 25.0, L: float = 25.0, phi_max: float = np.deg2rad(40.0)):
    """Surface mesh of a cylindrical roof segment -- the Scordelis-Lo quarter model of the reference's shell drivers
    (`examples/ongoing/shape_opt/run_shape_opt_roof.py:48-51,131-160`: axis along x in [0, L], y = R sin(phi), z = R cos(phi),
    phi in [0, phi_max]); every (x, phi) cell split into two triangles.  Returns (points (n, 3), triangles (m, 3) int64):
    the arguments of `ShellSpace` / `ShellProblem` (BASELINE config 3: nx = nphi = 362 gives 1.97 M dofs)."""
    xs = np.linspace(0.0, L, nx + 1)
    ph = np.linspace(0.0, phi_max, nphi + 1)
    X, P = np.meshgrid(xs, ph, indexing="ij")
    pts = np.stack([X.ravel(), R * np.sin(P).ravel(), R * np.cos(P).ravel()], axis=1)
    idx = np.arange((nx + 1) * (nphi + 1)).reshape(nx + 1, nphi + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])


def roof_quarter_model_dofs(space, L: float = 25.0):
    """Strongly imposed dofs of the Scordelis-Lo quarter model on a `ShellSpace` of `createCylindricalRoofMesh`
    (`run_shape_opt_roof.py:131-160`): rigid diaphragm at x = L (u_y = u_z = 0), symmetry planes x = 0 and y = 0."""
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    ux, vx = space.unode_x, space.x
    return np.unique(np.concatenate([
        space.u_dof(on(ux[:, 0], L), 1), space.u_dof(on(ux[:, 0], L), 2), space.u_dof(on(ux[:, 1], 0.0), 1), space.theta_dof(on(vx[:, 1], 0.0), 0),
        space.theta_dof(on(vx[:, 1], 0.0), 2), space.u_dof(on(ux[:, 0], 0.0), 0), space.theta_dof(on(vx[:, 0], 0.0), 1), space.theta_dof(on(vx[:, 0], 0.0), 2)]))


def createRectangleMesh(pt1, pt2, nx: int, ny: int) -> Mesh:
    """utils_dolfinx.py:148-153 creates quadrilaterals; the HIP engine is P1-simplex only, so every
    cell of the nx x ny grid over [pt1, pt2] is split along its right diagonal like create_unit_square."""
    (x0, y0), (x1, y1) = pt1, pt2
    gx = x0 + (x1 - x0) * np.arange(nx + 1) / nx
    gy = y0 + (y1 - y0) * np.arange(ny + 1) / ny
    gx[-1], gy[-1] = x1, y1
    x = np.empty(((nx + 1) * (ny + 1), 2))
    x[:, 0] = np.tile(gx, ny + 1)
    x[:, 1] = np.repeat(gy, nx + 1)
    jj, ii = np.divmod(np.arange(nx * ny), nx)
    v0 = jj * (nx + 1) + ii
    conn = np.empty((2 * nx * ny, 3), dtype=np.int32)
    conn[0::2, 0] = v0; conn[0::2, 1] = v0 + 1; conn[0::2, 2] = v0 + nx + 2
    conn[1::2, 0] = v0; conn[1::2, 1] = v0 + nx + 2; conn[1::2, 2] = v0 + nx + 1
    return Mesh(x, conn)


def meshSize(mesh: Mesh) -> np.ndarray:
    """utils_dolfinx.py:530-534 (dolfinx.cpp.mesh.h [ext]): per cell, the largest distance between two
    of its vertices."""
    p = mesh.x[mesh.conn]                                   # (n_cell, d+1, d)
    h = np.zeros(mesh.n_cell)
    d1 = mesh.conn.shape[1]
    for a in range(d1):
        for b in range(a + 1, d1):
            np.maximum(h, np.linalg.norm(p[:, a] - p[:, b], axis=1), out=h)
    return h


def findNodeIndices(node_coordinates, coordinates) -> np.ndarray:
    """utils_dolfinx.py:587-595: indices of the vertices of ``coordinates`` closest to the given points."""
    from scipy.spatial import KDTree
    _, idx = KDTree(np.asarray(coordinates)).query(np.asarray(node_coordinates))
    return idx


_KUHN = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


def createUnitCubeMesh(n: int, jitter: float = 0.0, seed: int = 20240807) -> Mesh:
    np1 = n + 1
    g = np.arange(np1) / n
    g[-1] = 1.0
    x = np.empty((np1 ** 3, 3))
    x[:, 0] = np.tile(g, np1 * np1)
    x[:, 1] = np.tile(np.repeat(g, np1), np1)
    x[:, 2] = np.repeat(g, np1 * np1)
    if np1 ** 3 >= 2 ** 31:
        raise ValueError("createUnitCubeMesh: vertex ids exceed int32")
    r = np.arange(n, dtype=np.int32)
    base = (r[:, None, None] * np.int32(np1 * np1) + r[None, :, None] * np.int32(np1) + r[None, None, :]).ravel()
    stride = (1, np1, np1 * np1)
    # vertex offsets of the six Kuhn tetrahedra of a cube: one broadcast add instead of 24 strided column writes
    # (14 s of the 16 s set-up at n = 215 were spent there)
    off = np.zeros((6, 4), dtype=np.int32)
    for t, perm in enumerate(_KUHN):
        acc = 0
        for s, ax in enumerate(perm):
            acc += stride[ax]
            off[t, s + 1] = acc
    conn = (base[:, None, None] + off[None, :, :]).reshape(-1, 4)
    return Mesh(_apply_jitter(x, n, jitter, seed), conn, n)


class BeamMesh(Mesh):
    """Interval mesh for the cubic-Hermite Euler-Bernoulli beam (utils_dolfinx.py:142-146
    createIntervalMesh -> dolfinx create_interval [ext]).  The engine sees the DOF graph: vertices
    are the DOFs (w_i, th_i), element e couples {2e, 2e+1, 2e+2, 2e+3} like a tetrahedron couples
    its vertices, so the tdim = 3 incidence / pattern / assembly walk is reused as is."""

    def __init__(self, n: int, x0: float, x1: float):
        self.nel = int(n)
        self.nodes = x0 + (x1 - x0) * np.arange(n + 1) / n
        self.nodes[-1] = x1
        x = np.zeros((2 * (n + 1), 3))
        x[:, 0] = np.repeat(self.nodes, 2)
        conn = (2 * np.arange(n)[:, None] + np.arange(4)[None, :]).astype(np.int32)
        super().__init__(x, conn, n)
        self.gdim = 1

    def cell_lengths(self) -> np.ndarray:
        return np.diff(self.nodes)

    def centroids(self) -> np.ndarray:
        return (0.5 * (self.nodes[1:] + self.nodes[:-1]))[:, None]

    def boundary_facet_mask(self):
        raise NotImplementedError("exterior facets of the beam are its two end points")


def createIntervalMesh(n: int, x0: float, x1: float) -> BeamMesh:
    """utils_dolfinx.py:142-146"""
    return BeamMesh(n, x0, x1)


def locate_dofs_geometrical(V, marker: Callable[[np.ndarray], np.ndarray]) -> np.ndarray:
    """dolfinx.fem.locate_dofs_geometrical [ext] for CG1: ``marker`` receives the
    coordinates as an array of shape (3, n_dofs) (gdim rows used, rest zero) and
    returns a boolean mask (run_poisson_opt.py:124-133)."""
    if isinstance(V, (tuple, list)):
        V = V[0]
    x = V.tabulate_dof_coordinates()
    xt = np.zeros((3, x.shape[0]))
    xt[:x.shape[1]] = x.T
    return np.nonzero(np.asarray(marker(xt), dtype=bool))[0].astype(np.int32)
