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
