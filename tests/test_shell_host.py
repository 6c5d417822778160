"""Host side of the shell's lattice preconditioner (femo_amd/fea/shell.py): the nested lattices and the plan of the
exact coarse solve (`femo_shell_pc_coarse`) -- no GPU."""
import numpy as np
import scipy.sparse as sp

from femo_amd.fea.shell import ShellSpace, coarse_solve_plan, lattice_pc
from oracle import shell_oracle as so


def test_lattice_levels_are_nested_partitions_of_unity():
    pts, conn = so.scordelis_lo_mesh(8, 8)
    S = ShellSpace(pts, conn)
    L = lattice_pc(S)
    assert L["levels"] == [2, 4, 8] and L["width"] == 24
    w = L["ell_w"].reshape(S.n_dof, len(L["levels"]), 8)
    assert np.allclose(w.sum(axis=2), 1.0) and w.min() >= 0.0
    # P_l = P_{l+1} T_l: the coarser level's interpolation is the finer one's composed with the node transfer
    off = L["level_offsets"]
    Tp = sp.csr_matrix((L["par_vals"], L["par_cols"], L["par_rowptr"]), shape=(L["n_nodes"], L["n_nodes"]))
    rows = np.repeat(np.arange(S.n_dof), 8)
    for l in range(len(L["levels"]) - 1):
        P = [sp.csr_matrix((L["ell_w"][:, 8 * k:8 * k + 8].ravel(), (rows, L["ell_idx"][:, 8 * k:8 * k + 8].ravel() // 6)),
                           shape=(S.n_dof, L["n_nodes"])) for k in (l, l + 1)]
        assert abs(P[0] - P[1] @ Tp).max() < 1e-14


def test_pattern_from_node_blocks_is_the_scalar_pattern():
    rng = np.random.default_rng(3)
    pts, conn = so.scordelis_lo_mesh(6, 5)
    conn = conn[rng.permutation(conn.shape[0])]
    conn = np.stack([np.roll(c, rng.integers(3)) for c in conn])              # any cell order, any rotation
    S = ShellSpace(pts, conn)
    for got, want in zip(S.pattern(), S.pattern_scalar_reference()):
        assert got.dtype == want.dtype and np.array_equal(got, want)


def test_coarse_solve_plan():
    pts, conn = so.scordelis_lo_mesh(16, 16)
    S = ShellSpace(pts, conn)
    L = lattice_pc(S)
    plan = coarse_solve_plan(L, chunk=64)
    c, off = plan["level"], L["level_offsets"]
    assert L["levels"][c] == 8 and 6 * (off[c + 1] - off[c]) <= 3200          # never the finest level (16)
    assert coarse_solve_plan(L, max_unknowns=10) is None
    n_pts = S.n_dof // 3
    ptr, ipts, nbr, xyz = plan["item_ptr"], plan["item_pts"], plan["item_nbr"], plan["node_xyz"]
    assert np.array_equal(np.sort(ipts), np.arange(n_pts))                    # every point in exactly one item
    assert np.all(np.diff(ptr) > 0) and np.diff(ptr).max() <= 64
    node = L["ell_idx"][0::3, 8 * c:8 * c + 8].astype(np.int64) // 6 - off[c]   # (n_pts, 8) level-local nodes of a point
    rowptr, cols, _ = S.pattern()
    for it in range(ptr.size - 1):
        p = ipts[ptr[it]:ptr[it + 1]]
        assert np.all(node[p, 0] == node[p[0], 0])                           # one coarse cell ...
        assert np.all((p >= S.n_unode) == (p[0] >= S.n_unode))               # ... and one field group per item
        base = xyz[node[p[0], 0]]
        # corner k of the cell sits at local (1 + k & 1, 1 + (k >> 1) & 1, 1 + (k >> 2) & 1) of the 4 x 4 x 4 table
        for k in range(8):
            loc = (1 + (k & 1)) + 4 * (1 + ((k >> 1) & 1)) + 16 * (1 + ((k >> 2) & 1))
            assert nbr[it, loc] == node[p[0], k]
        # every point any of the item's points couples to has its cell within one cell of the item's, and the
        # table holds its nodes where the kernel will look for them
        j = np.unique(np.concatenate([cols[rowptr[3 * i]:rowptr[3 * i + 1]:3] // 3 for i in p[:8]]))
        o = xyz[node[j, 0]] - base + 1
        assert o.min() >= 0 and o.max() <= 2
        for k in range(8):
            loc = (o[:, 0] + (k & 1)) + 4 * (o[:, 1] + ((k >> 1) & 1)) + 16 * (o[:, 2] + ((k >> 2) & 1))
            assert np.array_equal(nbr[it, loc], node[j, k])


def test_composite_restriction_is_the_chain_of_node_transfers():
    pts, conn = so.scordelis_lo_mesh(64, 64)
    L = lattice_pc(ShellSpace(pts, conn))
    plan = coarse_solve_plan(L)
    off, c, F, nn = L["level_offsets"], plan["level"], len(L["levels"]) - 1, L["n_nodes"]
    assert F - c >= 1
    Tc = sp.csr_matrix((L["chi_vals"], L["chi_cols"], L["chi_rowptr"]), shape=(nn, nn))
    g = np.zeros(nn)
    g[off[F]:off[F + 1]] = np.random.default_rng(0).standard_normal(off[F + 1] - off[F])
    chain = g.copy()
    for l in range(F - 1, c - 1, -1):
        chain[off[l]:off[l + 1]] = (Tc @ chain)[off[l]:off[l + 1]]
    D = sp.csr_matrix((plan["down_vals"], plan["down_cols"], plan["down_rowptr"]), shape=(off[F] - off[c], nn))
    assert plan["down_cols"].min() >= off[F]                                  # reads the finest lattice only
    assert np.abs(D @ g - chain[off[c]:off[F]]).max() <= 1e-14 * np.abs(g).max()


def test_roof_mesh_generator_is_the_oracles():
    """femo_amd/fea/mesh.py::createCylindricalRoofMesh (what bench.py's config-3 leg and the drivers use) against the
    oracle's Scordelis-Lo mesh: same points, same triangles; the quarter-model dof set against the tests' own."""
    from femo_amd.fea.mesh import createCylindricalRoofMesh, roof_quarter_model_dofs
    from femo_amd.fea.shell import ShellSpace
    from oracle import shell_oracle as so
    for nx, nphi in ((4, 4), (7, 5)):
        p, c = createCylindricalRoofMesh(nx, nphi)
        po, co = so.scordelis_lo_mesh(nx, nphi)
        assert np.array_equal(p, po) and np.array_equal(c, co)
    S = ShellSpace(p, c)
    V = so.ShellSpace(po, co)
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    ux, vx = V.unode_x, V.x
    ref = np.unique(np.concatenate([
        V.u_dof(on(ux[:, 0], 25.0), 1), V.u_dof(on(ux[:, 0], 25.0), 2), V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0),
        V.theta_dof(on(vx[:, 1], 0.0), 2), V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))
    assert np.array_equal(roof_quarter_model_dofs(S), ref)


def test_shell_pde_small_members():
    """ShellPDE.bf_sup_sizes, compute_alpha, compute_nodal_disp (shell_pde.py:233-244, 333-334) on a flat plate, where the
    answers are closed forms: vertex support sizes sum to the area, the projected cell diameter is the constant diameter,
    nodal displacements are the vertex part of the state."""
    from femo_amd.fea.shell_forms import ShellMesh, ShellPDE
    n, a = 6, 2.0
    g = np.linspace(0.0, a, n + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
    q = [idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()]
    conn = np.concatenate([np.stack([q[0], q[1], q[2]], axis=1), np.stack([q[0], q[2], q[3]], axis=1)])
    pde = ShellPDE(ShellMesh(pts, conn))
    sup = pde.bf_sup_sizes
    assert sup.shape == (pts.shape[0],) and abs(sup.sum() - a * a) < 1e-12
    h = a / n
    assert abs(sup[idx[2, 3]] - h * h) < 1e-12                      # interior vertex: six triangles of area h^2/2, a third each
    assert abs(pde.compute_alpha() - (np.sqrt(2.0) * h) ** 2 / 2.0) < 1e-10      # every cell diameter is the diagonal sqrt(2) h
    S = pde.mesh.space
    w = np.arange(S.n_dof, dtype=np.float64)
    ux, uy, uz = pde.compute_nodal_disp(w)
    assert np.array_equal(np.stack([ux, uy, uz], axis=1), S.vertex_displacement(w))


def test_node_block_items_group_points_by_cell():
    """`node_block_items` (plan of k_pc_galerkin_blocks_w): per level above the coarse solve every point appears once, the points of
    an item share their eight lattice nodes, items hold 1 .. 64 points, and the packed cell coordinates are those of the
    item's corner node."""
    from femo_amd.fea.mesh import createCylindricalRoofMesh
    from femo_amd.fea.shell import ShellSpace, coarse_solve_plan, lattice_pc, node_block_items
    x, conn = createCylindricalRoofMesh(12, 12)
    space = ShellSpace(np.asarray(x), np.asarray(conn))
    L = lattice_pc(space)
    c = coarse_solve_plan(L, 300)["level"]
    B = node_block_items(L, c + 1)
    n_pts = L["ell_idx"].shape[0] // 3
    n_above = len(L["levels"]) - 1 - c
    assert n_above >= 2 and B["pcell"].shape == (n_above, n_pts)
    ptr, lvl, pts = B["item_ptr"], B["item_lvl"], B["item_pts"]
    assert ptr[0] == 0 and ptr[-1] == n_above * n_pts and np.all(np.diff(ptr) >= 1) and np.all(np.diff(ptr) <= 64)
    for j in range(n_above):
        l = c + 1 + j
        sel = np.flatnonzero(lvl == j)
        allp = np.concatenate([pts[ptr[i]:ptr[i + 1]] for i in sel])
        assert np.array_equal(np.sort(allp), np.arange(n_pts))
        nodes = L["ell_idx"][0::3, 8 * l:8 * l + 8] // 6
        mm = L["levels"][l]
        for i in sel[:: max(1, sel.size // 40)]:
            p = pts[ptr[i]:ptr[i + 1]]
            assert np.all(nodes[p] == nodes[p[0]])
            g = L["level_nodes"][l][nodes[p[0], 0] - L["level_offsets"][l]]
            ck = B["pcell"][j, p[0]]
            assert (ck & 1023, (ck >> 10) & 1023, ck >> 20) == (g % (mm + 1), (g // (mm + 1)) % (mm + 1), g // (mm + 1) ** 2)
            assert np.all(B["pcell"][j, p] == ck)
