"""Mesh import (femo/fea/utils_dolfinx.py:69-123 import_mesh) and the XDMF reader behind it, without a GPU."""
import os

import numpy as np
import pytest

from tests.meshes import l_shape_mesh


@pytest.mark.parametrize("binary", [True, False])
def test_import_mesh_round_trip(tmp_path, binary):
    from femo_amd.fea.io import import_mesh, write_mesh_files
    from femo_amd.fea.mesh import Mesh
    x, conn, edges, tags = l_shape_mesh(6)
    mesh = Mesh(x, conn)
    cell_tags = (mesh.centroids()[:, 0] > 0).astype(np.int32) + 1
    table = {"outer": 1, "reentrant": 2, "left": 1, "right": 2}
    write_mesh_files("lshape", mesh, edges, tags, table, cell_tags=cell_tags, directory=str(tmp_path), binary=binary)
    m2, bmf, smf, table2 = import_mesh(prefix="lshape", subdomains=True, dim=2, directory=str(tmp_path), reorder=False)
    assert np.array_equal(m2.x, x) and np.array_equal(m2.conn, conn)           # bit for bit, repr() / raw binary
    assert np.array_equal(bmf.entities, edges) and np.array_equal(bmf.values, tags) and bmf.dim == 1
    assert np.array_equal(smf.values, cell_tags) and smf.dim == 2
    assert table2 == table
    # default: renumbered for locality like dolfinx's reader; the same mesh and tags through the index maps
    m3, bmf3, smf3, table3 = import_mesh(prefix="lshape", subdomains=True, dim=2, directory=str(tmp_path))
    assert m3.n_cell == mesh.n_cell and table3 == table
    assert np.array_equal(m3.x, x[m3.original_vertex_index])
    assert np.array_equal(m3.original_vertex_index[m3.conn], conn[m3.original_cell_index])
    assert np.array_equal(smf3.values, cell_tags[m3.original_cell_index])
    assert np.array_equal(np.sort(m3.original_vertex_index[bmf3.entities], axis=1), np.sort(edges, axis=1)) and np.array_equal(bmf3.values, tags)
    # locality: the average index distance between a cell's vertices shrinks by an order of magnitude
    spread = lambda c: np.abs(c.max(axis=1) - c.min(axis=1)).mean()
    assert spread(m3.conn) < 0.3 * spread(conn)
    # tagged vertices = what locate_dofs_topological gives for CG1
    v2 = bmf.vertices(2)
    assert np.all((np.abs(x[v2, 0]) < 1e-12) | (np.abs(x[v2, 1]) < 1e-12)) and len(v2) == 2 * 6 + 1
    assert len(np.union1d(bmf.vertices(1), v2)) == len(np.unique(edges))
    with pytest.raises(ValueError):
        import_mesh(prefix="lshape", dim=3, directory=str(tmp_path))


def test_reader_reads_the_recorders_files_and_refuses_hdf(tmp_path):
    from femo_amd.fea.io import XDMFRecorder, read_mesh, read_xdmf_grid
    from femo_amd.fea.mesh import createUnitCubeMesh
    mesh = createUnitCubeMesh(3, jitter=0.2)
    rec = XDMFRecorder(os.path.join(tmp_path, "record_u.xdmf"))
    rec.write_mesh(mesh)
    m2, _ = read_mesh(os.path.join(tmp_path, "record_u.xdmf"))
    assert np.array_equal(m2.x, mesh.x) and np.array_equal(m2.conn, mesh.conn)
    bad = os.path.join(tmp_path, "h5.xdmf")
    with open(bad, "w") as fh:
        fh.write('<Xdmf Version="3.0"><Domain><Grid Name="Grid" GridType="Uniform"><Topology TopologyType="Triangle" '
                 'NumberOfElements="1"><DataItem Format="HDF" DataType="Int" Dimensions="1 3">m.h5:/t</DataItem></Topology>'
                 '<Geometry GeometryType="XY"><DataItem Format="HDF" Dimensions="3 2">m.h5:/x</DataItem></Geometry></Grid></Domain></Xdmf>')
    with pytest.raises(NotImplementedError, match="HDF"):
        read_xdmf_grid(bad)


def test_l_shape_generator_is_a_valid_unstructured_mesh():
    x, conn, edges, tags = l_shape_mesh(10)
    e1, e2 = x[conn[:, 1]] - x[conn[:, 0]], x[conn[:, 2]] - x[conn[:, 0]]
    area = 0.5 * np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    assert area.min() > 0 and abs(area.sum() - 3.0) < 1e-12            # the L covers 3 of the 4 unit squares
    assert area.max() / area.min() > 20                                  # graded
    assert np.abs(np.diff(np.sort(conn.ravel()))).max() <= 1 and set(tags) == {1, 2}


def test_shell_mesh_round_trip(tmp_path):
    """Triangle grid with three coordinates per vertex (the roof meshes of run_shape_opt_roof.py:22-42), XML and raw."""
    from oracle import shell_oracle as so
    from femo_amd.fea.shell_forms import ShellMesh
    pts, conn = so.scordelis_lo_mesh(5, 3)
    mesh = ShellMesh(pts, conn)
    for binary in (False, True):
        path = str(tmp_path / f"roof_{int(binary)}.xdmf")
        mesh.write(path, binary=binary)
        back = ShellMesh.read(path)
        assert np.array_equal(back.conn, mesh.conn) and np.array_equal(back.x, mesh.x)
        assert back.space.n_dof == mesh.space.n_dof == 3 * (mesh.n_vert + mesh.space.n_edge) + 3 * mesh.n_vert
