"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol the header declares (no compute without a GPU), the host topology build, the
mesh generators, the form catalogue and the CSDL protocol stubs."""
import os
import re

import numpy as np
import pytest

from oracle import femo_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    # the product ABI (femo_hip.h) and the test-only entry points (femo_hip_test.h: rank emulation, model communicator)
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("femo_hip.h", "femo_hip_test.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(femo_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from femo_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/femo_hip.h but not exported"
    assert set(_lib.PROTOTYPES) == set(syms), set(_lib.PROTOTYPES) ^ set(syms)
    assert lib.femo_abi_version() == _lib.ABI_VERSION == 10


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to compute (it never routes to the oracle)."""
    from femo_amd import _lib
    from femo_amd.engine import Context
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.FemoError):
        Context(0)
    import femo_amd
    src = []
    for dp, _, fs in os.walk(os.path.dirname(femo_amd.__file__)):
        src += [open(os.path.join(dp, f)).read() for f in fs if f.endswith((".py", ".hip", ".cpp", ".h"))]
    assert not any(re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) for s in src)


@pytest.mark.parametrize("d,n,jit", [(2, 7, 0.0), (2, 33, 0.2), (3, 5, 0.0), (3, 17, 0.2)])
def test_mesh_generators_match_oracle(d, n, jit):
    from femo_amd.fea.mesh import createUnitCubeMesh, createUnitSquareMesh
    m = createUnitSquareMesh(n, jit) if d == 2 else createUnitCubeMesh(n, jit)
    o = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    assert np.array_equal(m.conn, o.conn)
    assert np.abs(m.x - o.x).max() < 1e-15
    assert m.conn.dtype == np.int32 and m.x.flags.c_contiguous


@pytest.mark.parametrize("d,n", [(2, 9), (3, 7), (3, 20)])
def test_host_topology_matches_oracle_pattern(d, n):
    from femo_amd.engine import topology_host
    m = fo.unit_square_mesh(n) if d == 2 else fo.unit_cube_mesh(n)
    info, rowptr, col = topology_host(d, m.n_vert, m.n_vert, m.conn)
    K = fo.stiffness(m)
    assert info["nnz"] == K.nnz and np.array_equal(rowptr, K.indptr) and np.array_equal(col, K.indices)
    assert info["max_rowlen"] == (6 if d == 2 else 14) and info["max_valence"] == (6 if d == 2 else 24)
    assert info["n_slices"] == -(-m.n_vert // 64) and info["sell_entries"] % 128 == 0


def test_host_topology_ragged_and_edge_cases():
    from femo_amd import _lib
    from femo_amd.engine import topology_host
    # shuffled vertex numbering of an unstructured-looking mesh: pattern must follow
    m = fo.unit_cube_mesh(4, jitter=0.2)
    perm = np.random.default_rng(1).permutation(m.n_vert)
    conn = perm[m.conn].astype(np.int32)
    info, rowptr, col = topology_host(3, m.n_vert, m.n_vert, conn)
    inv = np.argsort(perm)
    K = fo.stiffness(fo.OMesh(3, m.x[inv], conn))
    assert np.array_equal(rowptr, K.indptr) and np.array_equal(col, K.indices)
    # owned rows only (multi-GPU local mesh): rows [0, n_rows) of the full pattern
    n_rows = 37
    info2, rowptr2, col2 = topology_host(3, m.n_vert, n_rows, conn)
    assert np.array_equal(rowptr2, K.indptr[:n_rows + 1]) and np.array_equal(col2, K.indices[:K.indptr[n_rows]])
    # empty mesh and a single cell
    info3, rp3, _ = topology_host(2, 0, 0, np.zeros((0, 3), np.int32))
    assert info3["nnz"] == 0 and rp3.tolist() == [0]
    info4, rp4, col4 = topology_host(2, 3, 3, np.array([[0, 1, 2]], np.int32))
    assert rp4.tolist() == [0, 3, 6, 9] and col4.tolist() == [0, 1, 2] * 3
    # isolated vertex keeps a diagonal-only row; out-of-range connectivity is an error
    info5, rp5, col5 = topology_host(2, 4, 4, np.array([[0, 1, 2]], np.int32))
    assert rp5.tolist() == [0, 3, 6, 9, 10] and col5[-1] == 3
    with pytest.raises(_lib.FemoError):
        topology_host(2, 3, 3, np.array([[0, 1, 3]], np.int32))
    with pytest.raises(_lib.FemoError):
        topology_host(4, 3, 3, np.array([[0, 1, 2, 3, 4]], np.int32))


def test_csdl_protocol_stubs():
    from femo_amd.csdl_opt import _csdl_compat as cc
    if cc.HAVE_CSDL:
        pytest.skip("real csdl present")

    class Op(cc.CustomImplicitOperation):
        def initialize(self):
            self.parameters.declare('k', default=2)

        def define(self):
            self.add_input('a', shape=(3,))
            self.add_output('b', shape=(3,))
            self.declare_derivatives('*', '*')

    class M(cc.Model):
        def initialize(self):
            self.parameters.declare('n', types=int)

        def define(self):
            a = self.declare_variable('a', shape=(3,), val=1.0)
            out = cc.custom(a, op=Op(k=5))
            self.register_output('b', out)

    m = M(n=3)
    m.define()
    assert m.variables['b'].kind == 'output' and m.variables['b'].op.parameters['k'] == 5
    assert m.variables['b'].op.input_meta['a']['shape'] == (3,)
    with pytest.raises(TypeError):
        M(n='x')
    with pytest.raises(KeyError):
        M(zz=1)


def test_form_catalogue_is_closed():
    """Forms outside the catalogue are refused (module import needs no GPU)."""
    import femo_amd.fea.forms as forms

    class FakeSpace:
        family = "CG"

    class FakeFn:
        function_space = FakeSpace()

    u, f = FakeFn(), FakeFn()
    with pytest.raises(NotImplementedError):
        forms.PoissonResidual(u, f)            # f must be DG0
    f.function_space = type("S", (), {"family": "DG"})()
    r = forms.PoissonResidual(u, f)
    assert forms.derivative(r, u).rank == 2 and forms.derivative(r, f).wrt is f
    assert forms.derivative(r, FakeFn()).depends is False and forms.derivative(r, u).depends is True
    with pytest.raises(ValueError):
        forms.derivative(r, 3.0)
    with pytest.raises(NotImplementedError):
        forms.pdeRes(u, None, f, weak_bc=True)


@pytest.mark.parametrize("d,n,jit", [(2, 37, 0.2), (2, 64, 0.0), (3, 9, 0.25), (3, 21, 0.1), (3, 2, 0.0)])
def test_pc_plan_host_matches_the_oracle(d, n, jit):
    """femo_pc_plan_host (pc_plan.cpp, the host half of the BPX preconditioner) against the NumPy
    restatement: same lattice hierarchy, same packed coordinates, and a sort whose bricks and bins
    hold exactly the vertices that belong there."""
    from femo_amd import engine as E
    from oracle import bpx_oracle as bo
    from oracle import femo_oracle as fo
    m = fo.unit_square_mesh(n, jit) if d == 2 else fo.unit_cube_mesh(n, jit)
    P = E.pc_plan_host(m.x)
    lo, hi = m.x.min(axis=0), m.x.max(axis=0)
    bins, H = bo.choose_lattice(lo, hi, m.n_vert)
    assert P["levels"] == len(bins) and np.array_equal(P["bins"], np.array(bins))
    nF = bins[-1]
    b, t = bo._locate(m.x, lo, hi, nF, quantise=True)
    # 8 bytes per vertex: 2-D two words bin << 20 | fraction; 3-D one 64-bit word of three 21-bit fields bin << 12 | fraction
    pk = P["pk"]
    assert pk.shape == (m.n_vert, 2) and pk.dtype == np.uint32
    if d == 2:
        fields, bits = pk.astype(np.uint64), 20
    else:
        word = pk[:, 0].astype(np.uint64) | (pk[:, 1].astype(np.uint64) << np.uint64(32))
        fields = np.stack([(word >> np.uint64(21 * k)) & np.uint64((1 << 21) - 1) for k in range(3)], axis=1)
        bits = 12
        assert not np.any(word >> np.uint64(63))
    assert bits == bo.pk_bits(d)
    assert np.array_equal(fields >> np.uint64(bits), b.astype(np.uint64))
    assert np.array_equal((fields & np.uint64((1 << bits) - 1)).astype(np.float64) / (1 << bits), t)
    # the sort: a permutation; every brick / bin range holds exactly the vertices whose bin says so
    perm = P["perm"]
    assert np.array_equal(np.sort(perm), np.arange(m.n_vert))
    B = 4 if d == 3 else 8
    bs = b[perm]
    assert P["brick_ptr"][0] == 0 and P["brick_ptr"][-1] == m.n_vert and np.all(np.diff(P["brick_ptr"]) > 0)
    for k in range(P["n_bricks"]):
        lo_k, hi_k = P["brick_ptr"][k], P["brick_ptr"][k + 1]
        base = P["brick_base"][k][:d]
        inside = bs[lo_k:hi_k] - base
        assert inside.min() >= 0 and inside.max() < B
        local = inside[:, 0] + B * inside[:, 1] + (B * B * inside[:, 2] if d == 3 else 0)
        assert np.all(np.diff(local) >= 0)                       # sorted by bin inside the brick
        bp = P["bin_ptr"][k].astype(np.int64)
        assert bp[0] == 0 and bp[64] == hi_k - lo_k
        assert np.array_equal(np.searchsorted(local, np.arange(65)), bp)
        # stable: vertices of one bin stay in index order
        for q in np.unique(local)[:4]:
            seg = perm[lo_k + bp[q]: lo_k + bp[q + 1]]
            assert np.all(np.diff(seg) > 0)


def test_pc_plan_host_errors():
    from femo_amd import engine as E
    from femo_amd._lib import FemoError
    x = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0]])           # all on a line: degenerate box in y
    with pytest.raises(FemoError, match="degenerate bounding box"):
        E.pc_plan_host(x)
    # 3-D packed coordinates hold 9 bits of bin per axis (round 5: three 21-bit fields in one 64-bit word).  Round 6 (ADVICE
    # round 5): a mesh that would ask for 512 bins or more -- ~1e9 vertices -- gets the finest lattice that fits (384 bins, a
    # coarser mesh-to-lattice ratio) instead of an error; the NumPy restatement makes the same choice
    from oracle import bpx_oracle as bo
    g = np.linspace(0.0, 1.0, 4)
    x3 = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    for nvg in (5 * 10 ** 8, 2 * 10 ** 9, 10 ** 11):
        bins = E.pc_plan_host(x3, n_vert_global=nvg)["bins"]
        assert bins[-1].max() == 384
        ob, _ = bo.choose_lattice(np.zeros(3), np.ones(3), nvg)
        assert len(ob) == len(bins) and list(ob[-1]) == list(bins[-1])


def test_lattice_occupancy_flags_graded_meshes():
    """Mesh.lattice_occupancy(): ~1-3 on (jittered) uniform meshes, > BPX_MAX_OCCUPANCY on strongly
    graded ones, for which the solver layer keeps Jacobi (BPX has no levels below its finest lattice)."""
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import Mesh, createUnitCubeMesh, createUnitSquareMesh
    for mesh in (createUnitSquareMesh(48), createUnitSquareMesh(48, 0.25), createUnitCubeMesh(12), createUnitCubeMesh(12, 0.25)):
        assert 1.0 <= mesh.lattice_occupancy() <= 4.0
    base = createUnitCubeMesh(16)
    mild = Mesh(base.x ** 1.2, base.conn)
    hard = Mesh(base.x ** 2.0, base.conn)
    assert mild.lattice_occupancy() < utils_hip.BPX_MAX_OCCUPANCY < hard.lattice_occupancy()


def test_xdmf_recorder_files(tmp_path):
    """createRecorder's stand-in (fea_dolfinx.py:228-234): record_<name>.xdmf as an XDMF3 temporal
    collection whose binary side files hold exactly what was written."""
    import xml.etree.ElementTree as ET
    from femo_amd.fea.io import XDMFRecorder
    from femo_amd.fea.mesh import createUnitCubeMesh

    class Fn:
        def __init__(self, name, a):
            self.name, self._a = name, a
            self.vector = self

        def getArray(self):
            return self._a

    mesh = createUnitCubeMesh(3, 0.1)
    rec = XDMFRecorder(str(tmp_path / "records" / "record_u.xdmf"))
    rec.write_mesh(mesh)
    root = ET.parse(rec.path).getroot()
    assert root.tag == "Xdmf" and root.find("Domain/Grid/Topology").get("TopologyType") == "Tetrahedron"
    u0, u1, f = np.arange(mesh.n_vert, dtype=float), -np.arange(mesh.n_vert, dtype=float), np.linspace(0, 1, mesh.n_cell)
    rec.write_function(Fn("u", u0), 0)
    rec.write_function(Fn("u", u1), 1)
    rec.write_function(Fn("f", f), 1)
    root = ET.parse(rec.path).getroot()
    grids = root.findall("Domain/Grid/Grid")
    assert root.find("Domain/Grid").get("CollectionType") == "Temporal" and len(grids) == 3
    assert [g.find("Time").get("Value") for g in grids] == ["0.0", "1.0", "1.0"]
    centres = [g.find("Attribute").get("Center") for g in grids]
    assert centres == ["Node", "Node", "Cell"]
    d = os.path.dirname(rec.path)
    for g, ref in zip(grids, (u0, u1, f)):
        item = g.find("Attribute/DataItem")
        data = np.fromfile(os.path.join(d, item.text), dtype="<f8")
        assert int(item.get("Dimensions")) == data.size and np.array_equal(data, ref)
    geo = np.fromfile(os.path.join(d, grids[0].find("Geometry/DataItem").text), dtype="<f8").reshape(-1, 3)
    topo = np.fromfile(os.path.join(d, grids[0].find("Topology/DataItem").text), dtype="<i4").reshape(-1, 4)
    assert np.array_equal(geo, mesh.x) and np.array_equal(topo, mesh.conn)
    with pytest.raises(ValueError):
        rec.write_function(Fn("bad", np.zeros(5)), 2)


def test_small_mesh_utilities():
    """createRectangleMesh / meshSize / findNodeIndices (utils_dolfinx.py:148-153, 530-534, 587-595)."""
    from femo_amd.fea.mesh import createRectangleMesh, createUnitSquareMesh, findNodeIndices, meshSize
    m = createRectangleMesh((1.0, -1.0), (3.0, 0.0), 4, 2)
    assert m.n_vert == 15 and m.n_cell == 16
    assert np.allclose(m.x.min(axis=0), [1.0, -1.0]) and np.allclose(m.x.max(axis=0), [3.0, 0.0])
    om = fo.OMesh(2, m.x, m.conn)
    vol, _ = fo.cell_geometry(om)
    assert np.allclose(vol, 0.125) and abs(vol.sum() - 2.0) < 1e-14          # positively oriented, exact cover
    h = meshSize(m)
    assert h.shape == (16,) and np.allclose(h, np.hypot(0.5, 0.5))
    u = createUnitSquareMesh(8, 0.2)
    pts = u.x[[3, 40, 77]] + 1e-4
    assert list(findNodeIndices(pts, u.x)) == [3, 40, 77]



def test_abstract_fea_registry():
    """AbstractFEA (fea_dolfinx.py:20-67): plain registries, duplicate input names rejected, variadic
    argument lists kept as tuples."""
    from femo_amd.fea.fea_hip import AbstractFEA

    class Fn:
        def rename(self, a, b):
            self.name = a

    fea = AbstractFEA(mesh="m")
    f, u = Fn(), Fn()
    fea.add_input('f', f)
    with pytest.raises(ValueError, match="already been used"):
        fea.add_input('f', Fn())
    fea.add_state('u', u, 'R', 'f', 'g')
    fea.add_output('J', 'form', 'u', 'f')
    fea.add_strong_bc('bc0')
    assert f.name == 'f' and u.name == 'u' and fea.mesh == "m"
    assert fea.states_dict['u']['arguments'] == ('f', 'g') and fea.outputs_dict['J'] == {'form': 'form', 'arguments': ('u', 'f')}
    assert fea.bcs_list == ['bc0'] and set(fea.inputs_dict['f']) == {'function'}


def test_host_thread_pool_runs_and_lets_the_process_exit():
    """femo_host_copy / femo_host_axpby need no GPU; the pool's workers must not keep the interpreter
    from exiting (a static pool destroyed at exit blocked on its waiting workers)."""
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from femo_amd import engine as E\n"
        "x = np.random.default_rng(0).random(2_000_003); y = np.zeros_like(x)\n"
        "for _ in range(20): E.host_axpby(2.0, x, 0.0, y)\n"
        "z = np.full(x.size, np.nan); E.host_axpby(0.0, z, 0.0, z)\n"
        "w = np.empty_like(x); E.host_copy(w, x)\n"
        "assert np.array_equal(y, 2 * x) and not z.any() and np.array_equal(w, x)\n"
        "print('ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=root)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr


def test_regular_slices_are_completed_with_structural_zeros():
    """topology.cpp step 3.5: on a structured numbering every full slice whose completed columns stay inside [0, n_vert) is
    regular -- rows of boundary vertices get structural zeros at the deltas they lack -- while the exported CSR pattern
    stays the true one (the other topology tests compare it with the oracle's); a random numbering has none."""
    from femo_amd.engine import topology_host
    m = fo.unit_cube_mesh(20)
    info, rowptr, col = topology_host(3, m.n_vert, m.n_vert, m.conn)
    K = fo.stiffness(m)
    assert info["nnz"] == K.nnz and np.array_equal(rowptr, K.indptr) and np.array_equal(col, K.indices)
    n_full = m.n_vert // 64
    assert info["regular_slices"] >= n_full - 8 and info["max_rowlen"] == 14
    assert info["sell_entries"] <= 1.02 * 14 * 64 * info["n_slices"]
    perm = np.random.default_rng(0).permutation(m.n_vert)
    info2, _, _ = topology_host(3, m.n_vert, m.n_vert, perm[m.conn].astype(np.int32))
    assert info2["regular_slices"] == 0
    m2 = fo.unit_square_mesh(40)
    info3, rp3, col3 = topology_host(2, m2.n_vert, m2.n_vert, m2.conn)
    K2 = fo.stiffness(m2)
    assert np.array_equal(rp3, K2.indptr) and np.array_equal(col3, K2.indices) and info3["regular_slices"] >= m2.n_vert // 64 - 4
