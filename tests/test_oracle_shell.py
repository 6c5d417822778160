"""The Reissner-Mindlin shell oracle (oracle/shell_oracle.py): SURVEY.md section 8(f) row 3, BASELINE config 3.
CPU only -- the HIP path of this element is not built yet; these tests pin the formulation the
kernels will be checked against."""
import numpy as np
import pytest

from oracle import shell_oracle as so


def _roof_space(n=4, nu=0.3):
    pts, conn = so.scordelis_lo_mesh(n, n)
    V = so.ShellSpace(pts, conn)
    return V, so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, nu))


def test_space_and_symmetry():
    V, K = _roof_space()
    assert V.n_vert == 25 and V.n_edge == 56 and V.n_dof == 3 * (25 + 56) + 3 * 25
    assert abs(K - K.T).max() <= 1e-9 * abs(K).max()
    # every element matrix is positive semi-definite
    Ke = so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, 0.3)
    assert np.linalg.eigvalsh(Ke).min() >= -1e-9 * np.abs(Ke).max()


def test_rigid_body_motions_carry_no_energy():
    """Three translations and three rotations (u = om x x, theta = om) are in the null space: the drilling strain is
    written so that it vanishes for a rigid rotation about the normal."""
    V, K = _roof_space()
    scale = abs(K).max()
    for mode in range(6):
        w = np.zeros(V.n_dof)
        if mode < 3:
            w[mode:3 * V.n_unode:3] = 1.0
        else:
            om = np.eye(3)[mode - 3]
            w[:3 * V.n_unode] = np.cross(om, V.unode_x).ravel()
            w[3 * V.n_unode:] = np.tile(om, V.n_vert)
        assert np.abs(K @ w).max() <= 1e-12 * scale, mode
    # and nothing else is: exactly six zero eigenvalues
    ev = np.linalg.eigvalsh(K.toarray())
    assert np.count_nonzero(ev < 1e-9 * ev.max()) == 6


def test_one_point_shear_rule_is_rank_deficient():
    """Why dx_shear is the 3-point rule: with one point per element the supported roof is singular."""
    pts, conn = so.scordelis_lo_mesh(4, 4)
    V = so.ShellSpace(pts, conn)
    h = np.full(V.n_vert, 0.25)
    ev3 = np.linalg.eigvalsh(so.assemble(V, so.element_stiffness(V, h, 4.32e8, 0.0)).toarray())
    ev1 = np.linalg.eigvalsh(so.assemble(V, so.element_stiffness(V, h, 4.32e8, 0.0, quad_shear=so.QUAD_ONE_POINT)).toarray())
    assert np.count_nonzero(ev3 < 1e-9 * ev3.max()) == 6
    assert np.count_nonzero(ev1 < 1e-9 * ev1.max()) > 6


def test_uniform_stretch_patch():
    """u = eps x e_x on a flat plate: membrane energy 1/2 h E/(1-nu^2) eps^2 A, every other part zero."""
    pts, conn = so.plate_mesh(3)
    V = so.ShellSpace(pts, conn)
    E, nu, h, eps = 2.0e5, 0.3, 0.02, 1e-3
    w = np.zeros(V.n_dof)
    w[0:3 * V.n_unode:3] = eps * V.unode_x[:, 0]
    parts = so.energy_parts(V, w, np.full(V.n_vert, h), E, nu)
    assert parts["membrane"] == pytest.approx(0.5 * h * E / (1 - nu ** 2) * eps ** 2, rel=1e-12)
    assert max(parts["bending"], parts["shear"], parts["drilling"]) <= 1e-20


def test_thickness_enters_as_h_and_h_cubed():
    V, _ = _roof_space()
    km, kb, ks, kd = so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, 0.3, return_parts=True)
    km2, kb2, ks2, kd2 = so.element_stiffness(V, np.full(V.n_vert, 0.5), 4.32e8, 0.3, return_parts=True)
    assert np.allclose(km2, 2 * km) and np.allclose(ks2, 2 * ks) and np.allclose(kb2, 8 * kb) and np.allclose(kd2, 8 * kd)


def test_simply_supported_plate_matches_kirchhoff():
    # thin plate (h / a = 0.01): the CG2/CG1 pair locks on coarse meshes (0.29, 0.78 of the answer at n = 4, 8)
    # and converges: 0.979 at n = 16, 1.0008 at n = 32 (the shear-deformable answer lies slightly above Kirchhoff's)
    w16, ref = so.simply_supported_plate(16)
    w32, _ = so.simply_supported_plate(32)
    assert w16 == pytest.approx(ref, rel=0.03) and w32 == pytest.approx(ref, rel=0.003)
    w8, _ = so.simply_supported_plate(8)
    assert abs(w8 - ref) > abs(w16 - ref) > abs(w32 - ref)


def test_scordelis_lo_roof():
    """The one shell number the reference tree holds: v_tip = -0.3024 (run_shape_opt_roof.py:224), approached from
    below like every displacement-based element."""
    t8, _, _ = so.scordelis_lo(8, 8)
    t16, V, w = so.scordelis_lo(16, 16)
    assert -0.3024 < t16 < t8 < 0.0
    assert t16 == pytest.approx(-0.3024, rel=0.015)
    parts = so.energy_parts(V, w, np.full(V.n_vert, 0.25), 4.32e8, 0.0)
    assert parts["shear"] < 0.01 * (parts["membrane"] + parts["bending"])      # thin: no shear locking
    assert so.compliance(V, w) > 0.0


def test_energy_does_not_depend_on_cell_orientation_or_numbering():
    """Flipping the vertex order of some triangles flips their normal and tangent frame; rotating the local numbering
    changes the frame's first axis.  The assembled stiffness must not care (the material is isotropic, every strain
    enters quadratically)."""
    pts, conn = so.scordelis_lo_mesh(4, 4)
    rng = np.random.default_rng(0)
    V0 = so.ShellSpace(pts, conn)
    h = 0.25 * (1 + 0.3 * rng.random(V0.n_vert))
    K0 = so.assemble(V0, so.element_stiffness(V0, h, 4.32e8, 0.3))
    conn2 = conn.copy()
    flip = rng.random(len(conn)) < 0.5
    conn2[flip] = conn2[flip][:, [0, 2, 1]]
    rot = rng.integers(0, 3, len(conn))
    conn2 = np.stack([np.roll(c, r) for c, r in zip(conn2, rot)])
    V1 = so.ShellSpace(pts, conn2)
    assert V1.n_dof == V0.n_dof and np.array_equal(V1.edge_vertices, V0.edge_vertices)      # same global numbering
    K1 = so.assemble(V1, so.element_stiffness(V1, h, 4.32e8, 0.3))
    assert abs(K1 - K0).max() <= 1e-10 * abs(K0).max()


def test_plate_bending_stress_matches_the_series_solution():
    """Surface stress at the centre of a simply supported square plate under uniform pressure: sigma_max =
    6 M / h^2 with M = 0.0479 q a^2 for nu = 0.3 (Timoshenko & Woinowsky-Krieger, table 8 [ext]); there sigma_11 =
    sigma_22 and sigma_12 = 0, so the von Mises stress is sigma_max.  Pins `von_mises_stress` (shell_pde.py:315-328)."""
    n, h, E, nu, q = 16, 0.01, 1.0e7, 0.3, -1.0
    pts, conn = so.plate_mesh(n)
    V = so.ShellSpace(pts, conn)
    K = so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, h), E, nu))
    F = so.load_vector(V, np.tile([0.0, 0.0, q], (V.n_vert, 1)))
    ux = V.unode_x
    edge = np.nonzero(np.isclose(ux[:, 0], 0) | np.isclose(ux[:, 0], 1) | np.isclose(ux[:, 1], 0) | np.isclose(ux[:, 1], 1))[0]
    fixed = np.concatenate([V.u_dof(edge, 2), V.u_dof(np.arange(V.n_unode), 0), V.u_dof(np.arange(V.n_unode), 1),
                            V.theta_dof(np.arange(V.n_vert), 2)])
    w = so.solve(K, F, fixed)
    hn = np.full(V.n_vert, h)
    top, mid, bot = (so.von_mises_stress(V, w, hn, E, nu, s) for s in (1.0, 0.0, -1.0))
    cells = np.nonzero(np.all(np.abs(V.x[V.conn][:, :, :2] - 0.5).max(axis=2) <= 1.5 / n, axis=1))[0]    # around the centre
    ref = 6.0 * 0.0479 * abs(q) / h ** 2
    assert top[cells].max() == pytest.approx(ref, rel=0.02)
    assert np.allclose(top, bot, rtol=1e-9, atol=1e-9 * ref)          # pure bending: symmetric about the mid-surface
    assert mid.max() <= 1e-6 * ref                                    # no membrane stress


def test_pnorm_stress_partials():
    """dJ/dw and dJ/dh of the aggregated stress (shell_pde.py:297-313) against central differences."""
    pts, conn = so.scordelis_lo_mesh(4, 3)
    V = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(11)
    w = 1e-3 * rng.standard_normal(V.n_dof)
    h = 0.25 * (1.0 + 0.2 * rng.random(V.n_vert))
    E, nu, m, rho = 4.32e8, 0.3, 2e-6, 6.0
    for surface in (1.0, -1.0, 0.0):
        J, gw, gh = so.pnorm_stress(V, w, h, E, nu, m=m, rho=rho, surface=surface, grad=True)
        assert J > 0.0
        dw, dh = 1e-3 * rng.standard_normal(V.n_dof), 0.25 * rng.standard_normal(V.n_vert)
        f = lambda t, s: so.pnorm_stress(V, w + t * dw, h + s * dh, E, nu, m=m, rho=rho, surface=surface)
        assert gw @ dw == pytest.approx((f(1e-4, 0) - f(-1e-4, 0)) / 2e-4, rel=1e-6)
        assert gh @ dh == pytest.approx((f(0, 1e-5) - f(0, -1e-5)) / 2e-5, rel=1e-6, abs=1e-12 * abs(J))
    # alpha defaults to the area; rho = 1, m = 1: the mean von Mises stress
    _, _, _, area, _ = V.frames()
    vm = so.von_mises_stress(V, w, h, E, nu, 1.0)
    wq = np.asarray(so.QUAD_INPLANE[1])
    assert so.pnorm_stress(V, w, h, E, nu, m=1.0, rho=1.0) == pytest.approx(float((area[:, None] * wq[None, :] * vm).sum() / area.sum()), rel=1e-13)


def test_projection_of_the_von_mises_stress():
    """`project_von_mises` (shell_pde.py:330-332): the consistent projection reproduces a stress field that is P1 on the
    mesh -- pure bending of the plate with a linearly varying thickness gives sigma_vm = c h(x) -- and the lumped one keeps
    its integral."""
    pts, conn = so.plate_mesh(6)
    V = so.ShellSpace(pts, conn)
    E, nu = 1.0e7, 0.3
    # rotation field theta = (0, kappa x, 0): curvature kappa_11 constant, no membrane strain -> sigma = C (z kappa)
    w = np.zeros(V.n_dof)
    kap = 1e-3
    w[V.theta_dof(np.arange(V.n_vert), 1)] = kap * V.x[:, 0]
    h = 0.01 * (1.0 + 0.5 * V.x[:, 0] + 0.25 * V.x[:, 1])
    vm_q = so.von_mises_stress(V, w, h, E, nu, 1.0)
    C = so.plane_stress(E, nu)
    s = C @ np.array([kap, 0.0, 0.0])
    c = float(np.sqrt(s[0] ** 2 - s[0] * s[1] + s[1] ** 2)) / 2.0          # sigma_vm = c h
    lamq = np.asarray(so.QUAD_INPLANE[0])
    assert np.allclose(vm_q, c * np.einsum("cv,qv->cq", h[V.conn], lamq), rtol=1e-10)
    x = so.project_von_mises(V, w, h, E, nu, 1.0)
    assert np.allclose(x, c * h, rtol=1e-10)
    xl = so.project_von_mises(V, w, h, E, nu, 1.0, lump_mass=True)
    _, _, _, area, _ = V.frames()
    nodal = np.zeros(V.n_vert); np.add.at(nodal, V.conn.ravel(), np.repeat(area / 3.0, 3))
    assert float(nodal @ xl) == pytest.approx(float(nodal @ x), rel=1e-12)
    assert not np.allclose(xl, x, rtol=1e-4)


# ---------------------------------------------------------------------------------------------- round 3
def _small_roof():
    pts, conn = so.scordelis_lo_mesh(6, 6)
    V = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(0)
    return V, rng, 0.25 * (1.0 + 0.3 * rng.random(V.n_vert))


def test_exact_thickness_derivative_of_the_bilinear_form():
    V, rng, h = _small_roof()
    v, w = rng.standard_normal(V.n_dof), rng.standard_normal(V.n_dof)
    E, nu = 4.32e8, 0.3
    g = so.dform_dh(V, h, E, nu, v, w)
    dh = 0.01 * rng.standard_normal(V.n_vert)
    form = lambda hh: float(np.einsum("ca,cab,cb->", v[V.cell_dofs], so.element_stiffness(V, hh, E, nu), w[V.cell_dofs]))
    # K is cubic in h: Richardson-extrapolated central differences are exact up to round-off
    d1 = (form(h + 1e-2 * dh) - form(h - 1e-2 * dh)) / 2e-2
    d2 = (form(h + 2e-2 * dh) - form(h - 2e-2 * dh)) / 4e-2
    assert g @ dh == pytest.approx((4 * d1 - d2) / 3, rel=1e-10)
    # homogeneity: h d/dh of (membrane + shear) + (bending + drilling) parts = 1 x and 3 x the parts
    parts = so.element_stiffness(V, h, E, nu, return_parts=True)
    val = [float(np.einsum("ca,cab,cb->", v[V.cell_dofs], P, w[V.cell_dofs])) for P in parts]
    assert g @ h == pytest.approx(val[0] + 3 * val[1] + val[2] + 3 * val[3], rel=1e-11)


def test_compliance_on_a_tagged_subset_and_its_gradient():
    V, rng, _ = _small_roof()
    w, d = rng.standard_normal(V.n_dof), rng.standard_normal(V.n_dof)
    cells = np.arange(0, V.conn.shape[0], 3)
    rest = np.setdiff1d(np.arange(V.conn.shape[0]), cells)
    assert so.compliance(V, w) == pytest.approx(so.compliance(V, w, cells) + so.compliance(V, w, rest), rel=1e-13)
    g = so.compliance_du(V, w, cells)
    assert np.all(g[3 * V.n_unode:] == 0.0)
    assert g @ d == pytest.approx((so.compliance(V, w + d, cells) - so.compliance(V, w - d, cells)) / 2.0, rel=1e-12)   # quadratic: exact
    one = np.zeros(V.n_dof); one[0:3 * V.n_unode:3] = 1.0
    _, _, _, area, _ = V.frames()
    assert 2.0 * so.compliance(V, one) == pytest.approx(area.sum(), rel=1e-13)


def test_regularisation_terms():
    """shell_pde.py:262-282: values on fields with closed forms, gradients against differences."""
    pts, conn = so.plate_mesh(6, a=2.0)
    V = so.ShellSpace(pts, conn)
    hlin = 0.1 + 0.3 * V.x[:, 0] - 0.2 * V.x[:, 1]                      # |grad h|^2 = 0.13 on an area of 4
    assert so.regularization(V, hlin, "H1") == pytest.approx(0.5 * 1e3 * 0.13 * 4.0, rel=1e-12)
    assert so.regularization(V, np.full(V.n_vert, 0.5), "L2") == pytest.approx(0.5 * 1e3 * 0.25 * 4.0, rel=1e-12)
    hE = so.cell_diameter(V)
    assert np.allclose(hE, np.sqrt(2.0) * 2.0 / 6)
    assert so.regularization(V, hlin, "L2H1") == pytest.approx(so.regularization(V, hlin, "L2") + 0.5 * hE[0] ** 2 * 0.13 * 4.0, rel=1e-12)
    assert so.regularization(V, hlin, None) == 0.0
    rng = np.random.default_rng(3)
    h, dh = 0.2 + 0.1 * rng.random(V.n_vert), rng.standard_normal(V.n_vert)
    for kind in ("H1", "L2", "L2H1"):
        _, g = so.regularization(V, h, kind, grad=True)
        assert g @ dh == pytest.approx((so.regularization(V, h + dh, kind) - so.regularization(V, h - dh, kind)) / 2.0, rel=1e-11)
    with pytest.raises(ValueError):
        so.regularization(V, h, "H2")


def test_penalty_boundary_terms_reach_the_strong_limit():
    """The penalty form of `pdeRes(..., penalty=True, dss, dSS, g)`: symmetric positive semi-definite, acts on the tagged
    edges only, and beta -> 1e15 (the value the reference's recorded runs name) reproduces the strongly clamped solution."""
    import scipy.sparse.linalg as spla
    pp, cc = so.plate_mesh(8)
    P = so.ShellSpace(pp, cc)
    ext, inte = so.tagged_edges(P, lambda x: x[0] <= 1e-9)
    assert len(ext) == 8 and len(inte) == 0
    ext2, int2 = so.tagged_edges(P, lambda x: x[0] <= 0.25 + 1e-9)          # a clamped REGION: interior facets too (dS)
    assert len(ext2) == 8 + 2 * 2 and len(int2) > 0
    K = so.assemble(P, so.element_stiffness(P, np.full(P.n_vert, 0.05), 1e7, 0.3))
    F = so.load_vector(P, np.tile([0.0, 0.0, -1.0], (P.n_vert, 1)))
    un, vn = np.nonzero(P.unode_x[:, 0] <= 1e-9)[0], np.nonzero(P.x[:, 0] <= 1e-9)[0]
    fixed = np.concatenate([P.u_dof(un, k) for k in range(3)] + [P.theta_dof(vn, k) for k in range(3)])
    ws = so.solve(K, F, fixed)
    err = {}
    for beta in (1e6, 1e10, 1e15):
        Kp = so.penalty_matrix(P, ext, inte, beta)
        assert abs(Kp - Kp.T).max() == 0.0
        touched = np.unique(Kp.nonzero()[0])
        assert np.array_equal(touched, np.sort(fixed))
        err[beta] = np.abs(spla.spsolve((K + Kp).tocsc(), F) - ws).max() / np.abs(ws).max()
    assert err[1e6] > err[1e10] > err[1e15] and err[1e15] < 1e-9 and err[1e10] < 1e-5
    # edge mass matrices integrate constants exactly: 1^T K_pen 1 over one field = beta sum(len / h_E)
    Kp = so.penalty_matrix(P, ext, inte, 2.0)
    one = np.zeros(P.n_dof); one[0:3 * P.n_unode:3] = 1.0
    assert one @ (Kp @ one) == pytest.approx(2.0 * 8 * (1.0 / 8) / so.cell_diameter(P)[0], rel=1e-12)
    # inhomogeneous data: residual K_pen (w - g) vanishes at w = g
    g = np.random.default_rng(1).standard_normal(P.n_dof)
    assert np.abs(Kp @ (g - g)).max() == 0.0


def test_inertial_residual():
    """kinetic_residual (shell_pde.py:255-256): symmetric, total mass and rotary inertia of a uniform plate."""
    pts, conn = so.plate_mesh(4, a=2.0)
    V = so.ShellSpace(pts, conn)
    h, rho = np.full(V.n_vert, 0.1), 3.0
    rng = np.random.default_rng(2)
    a, b = rng.standard_normal(V.n_dof), rng.standard_normal(V.n_dof)
    assert a @ so.inertia_apply(V, h, rho, b) == pytest.approx(b @ so.inertia_apply(V, h, rho, a), rel=1e-12)
    tz = np.zeros(V.n_dof); tz[2:3 * V.n_unode:3] = 1.0                     # unit acceleration in z
    assert tz @ so.inertia_apply(V, h, rho, tz) == pytest.approx(rho * 0.1 * 4.0, rel=1e-12)
    rx = np.zeros(V.n_dof); rx[3 * V.n_unode::3] = 1.0                      # unit angular acceleration about x
    assert rx @ so.inertia_apply(V, h, rho, rx) == pytest.approx(rho * 0.1 ** 3 / 12.0 * 4.0, rel=1e-12)
    hv = 0.1 * (1.0 + V.x[:, 0])                                            # linear thickness: int rho h = rho 0.1 (4 + 4)
    assert tz @ so.inertia_apply(V, hv, rho, tz) == pytest.approx(rho * 0.1 * 8.0, rel=1e-12)


def test_reference_cycle_of_the_roof():
    out = so.reference_cycle(8)
    assert out["tip"] == pytest.approx(so.scordelis_lo(8, 8)[0], rel=1e-9)          # three Newton steps on a linear problem = one solve
    V = so.ShellSpace(*so.scordelis_lo_mesh(8, 8))
    assert out["J"] == pytest.approx(so.compliance(V, out["w"]), rel=1e-13)
    assert out["grad"].shape == (V.n_vert,) and np.all(np.isfinite(out["grad"]))


def _roof_problem(n):
    pts, conn = so.scordelis_lo_mesh(n, n)
    V = so.ShellSpace(pts, conn)
    K = so.assemble(V, so.element_stiffness(V, np.full(V.n_vert, 0.25), 4.32e8, 0.0)).tocsr()
    F = so.load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1)))
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    ux, vx = V.unode_x, V.x
    fixed = np.unique(np.concatenate([
        V.u_dof(on(ux[:, 0], 25.0), 1), V.u_dof(on(ux[:, 0], 25.0), 2), V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0),
        V.theta_dof(on(vx[:, 1], 0.0), 2), V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))
    return pts, conn, V, K, F, fixed


def test_lattice_preconditioner_hermite_spaces_halve_the_iteration_count():
    """Round 4: the lattice preconditioner of the shell's CG solves restated in SciPy (LatticePreconditioner) with trilinear
    lattice spaces (rounds 2-3) and with the Hermite-type ones (rotations as slopes).  Pinned: the operator is symmetric
    positive definite, PCG reaches the direct solution, and the iteration counts -- 16 x 16 roof 258 -> 153, 32 x 32 roof
    149 -> 85 with the node-block levels weighted 0.3 (the GPU reproduces them, tests/test_gpu_shell_hermite.py)."""
    counts = {}
    for n in (16, 32):
        pts, conn, V, K, F, fixed = _roof_problem(n)
        w_ref = so.solve(K, F, fixed)
        for herm in (False, True):
            M = so.LatticePreconditioner(V, K, fixed, hermite=herm)
            x, its = M.pcg(F)
            assert np.abs(x - w_ref).max() <= 1e-9 * np.abs(w_ref).max()
            counts[(n, herm)] = its
            if n == 16:
                rng = np.random.default_rng(1)
                a, b = rng.standard_normal(V.n_dof) * M.mask, rng.standard_normal(V.n_dof) * M.mask
                za, zb = M.apply(a), M.apply(b)
                assert abs(a @ zb - za @ b) <= 1e-9 * abs(a @ zb) and a @ za > 0.0 and b @ zb > 0.0
    assert abs(counts[(16, False)] - 258) <= 3 and abs(counts[(16, True)] - 153) <= 3
    assert abs(counts[(32, False)] - 149) <= 3 and abs(counts[(32, True)] - 85) <= 3
    # level_weight = 1 (the plain sum of rounds 2-4a): 269 / 156 and 168 / 89
    pts, conn, V, K, F, fixed = _roof_problem(32)
    assert abs(so.LatticePreconditioner(V, K, fixed, hermite=True, level_weight=1.0).pcg(F)[1] - 89) <= 3


def test_host_hermite_weights_are_the_oracles_composed_prolongations():
    """femo_amd/fea/shell.py::hermite_lattice composes the per-level (alpha, sigma) weights in closed form (separable 2 x 2
    contractions per axis); the oracle multiplies sparse matrices.  Entry by entry the same prolongations on every level,
    and the (a, b, c) transfer entries reproduce the oracle's lattice transfers."""
    import scipy.sparse as sp
    from femo_amd.fea.shell import ShellSpace, hermite_lattice, hermite_transfer_matrix, lattice_pc
    pts, conn, V, K, F, fixed = _roof_problem(16)
    S = ShellSpace(pts, conn)
    L = lattice_pc(S)
    H = hermite_lattice(S, L)
    M = so.LatticePreconditioner(V, K, [], hermite=True)           # no fixed dofs: the prolongations themselves
    assert M.levels == L["levels"]
    nl, nu, off = len(M.levels), S.n_unode, L["level_offsets"]
    for l in range(nl):
        assert np.array_equal(M.sets[l], L["level_nodes"][l])
        w4 = H["fin_w4"] if l == nl - 1 else H["lvl_w4"][l]
        idx = L["ell_idx"][0::3, 8 * l:8 * l + 8].astype(np.int64) // 6 - off[l]
        R, C, Vv = [], [], []
        p = np.arange(idx.shape[0])
        isu = p < nu
        for a in range(8):
            for i in range(3):
                R.append(3 * p + i); C.append(np.where(isu, 6 * idx[:, a] + i, 6 * idx[:, a] + 3 + i)); Vv.append(w4[:, a, 0].astype(float))
            for (i, k, j, sg) in ((0, 1, 2, 1.), (0, 2, 1, -1.), (1, 2, 0, 1.), (1, 0, 2, -1.), (2, 0, 1, 1.), (2, 1, 0, -1.)):
                R.append(3 * p[isu] + i); C.append(6 * idx[isu, a] + 3 + k); Vv.append(sg * w4[isu, a, 1 + j].astype(float))
        P = sp.csr_matrix((np.concatenate(Vv), (np.concatenate(R), np.concatenate(C))), shape=(S.n_dof, 6 * M.sets[l].size))
        assert abs(P - M.P[l]).max() <= 2e-7 * abs(M.P[l]).max()              # packed as float32
    for l in range(nl - 1):
        T = hermite_transfer_matrix(L, H, l)[6 * off[l + 1]:6 * off[l + 2], 6 * off[l]:6 * off[l + 1]]
        assert abs(T - M.T[l]).max() <= 1e-14
