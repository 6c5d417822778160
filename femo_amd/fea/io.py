"""XDMF time-series recorder (fea_dolfinx.py:228-234 createRecorder; dolfinx.io.XDMFFile [ext]).

The reference writes ``record_<name>.xdmf`` + HDF5 through dolfinx; h5py is not part of this
stack, so the heavy data go to raw little-endian binary side files that XDMF3 readers (ParaView)
open through ``Format="Binary"`` DataItems: same file name, same calls (``write_mesh``,
``write_function(function, t)``), one temporal collection with a grid per recorded iteration.
CG1 functions are written as node data, DG0 functions as cell data."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np

_TOPOLOGY = {2: ("Triangle", 3), 3: ("Tetrahedron", 4)}


class XDMFRecorder:
    def __init__(self, path: str):
        self.path = path if path.endswith(".xdmf") else path + ".xdmf"
        self.stem = self.path[:-5]
        os.makedirs(os.path.dirname(self.path) or ".", exist_ok=True)
        self._mesh = None
        self._steps: List[Tuple[float, str, str, str, int]] = []      # (time, field name, file, centre, count)

    # ------------------------------------------------------------------ API ----
    def write_mesh(self, mesh) -> None:
        if mesh.tdim not in _TOPOLOGY or mesh.conn.shape[1] != mesh.tdim + 1:
            raise NotImplementedError("XDMF recorder: triangle / tetrahedron meshes only")
        x = np.zeros((mesh.n_vert, 3), dtype="<f8")
        x[:, :mesh.tdim] = mesh.x
        x.tofile(self.stem + "_geometry.bin")
        np.ascontiguousarray(mesh.conn, dtype="<i4").tofile(self.stem + "_topology.bin")
        self._mesh = (mesh.tdim, mesh.n_vert, mesh.n_cell)
        self._flush()

    def write_function(self, function, t=0) -> None:
        if self._mesh is None:
            raise RuntimeError("write_mesh must be called before write_function")
        a = np.ascontiguousarray(function.vector.getArray(), dtype="<f8")
        _, n_vert, n_cell = self._mesh
        if a.size == n_vert:
            centre = "Node"
        elif a.size == n_cell:
            centre = "Cell"
        else:
            raise ValueError(f"function of size {a.size} matches neither the vertices ({n_vert}) nor the cells ({n_cell})")
        fname = f"{self.stem}_{len(self._steps):05d}.bin"
        a.tofile(fname)
        self._steps.append((float(t), getattr(function, "name", None) or "f", os.path.basename(fname), centre, a.size))
        self._flush()

    def close(self) -> None:
        self._flush()

    # ------------------------------------------------------------- internals ----
    def _flush(self) -> None:
        tdim, n_vert, n_cell = self._mesh
        kind, npc = _TOPOLOGY[tdim]
        base = os.path.basename(self.stem)
        geo = (f'<Geometry GeometryType="XYZ"><DataItem Format="Binary" DataType="Float" Precision="8" Endian="Little" '
               f'Dimensions="{n_vert} 3">{base}_geometry.bin</DataItem></Geometry>')
        topo = (f'<Topology TopologyType="{kind}" NumberOfElements="{n_cell}"><DataItem Format="Binary" DataType="Int" '
                f'Precision="4" Endian="Little" Dimensions="{n_cell} {npc}">{base}_topology.bin</DataItem></Topology>')
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', ' <Domain>']
        if not self._steps:
            out.append(f'  <Grid Name="mesh" GridType="Uniform">{topo}{geo}</Grid>')
        else:
            out.append('  <Grid Name="TimeSeries" GridType="Collection" CollectionType="Temporal">')
            for t, name, fname, centre, count in self._steps:
                out.append(f'   <Grid Name="{name}" GridType="Uniform"><Time Value="{t!r}"/>{topo}{geo}'
                           f'<Attribute Name="{name}" AttributeType="Scalar" Center="{centre}"><DataItem Format="Binary" '
                           f'DataType="Float" Precision="8" Endian="Little" Dimensions="{count}">{fname}</DataItem>'
                           f'</Attribute></Grid>')
            out.append('  </Grid>')
        out += [' </Domain>', '</Xdmf>']
        with open(self.path, "w") as fh:
            fh.write("\n".join(out) + "\n")


# ------------------------------------------------------------------ mesh import ----
# femo/fea/utils_dolfinx.py:69-123 import_mesh: reads `<prefix>_domain.xdmf` (the cells, optionally with subdomain
# tags), `<prefix>_boundaries.xdmf` (tagged boundary facets) and `<prefix>_association_table.ini`, the files the
# msh2xdmf converter leaves next to a gmsh mesh.  dolfinx's XDMFFile reads HDF5 heavy data; h5py is not part of
# this stack, so the reader takes XDMF files whose DataItems are inline (Format="XML", what meshio writes with
# data_format="XML") or raw binary (Format="Binary", what XDMFRecorder and write_mesh_files below write) and says
# so when it meets Format="HDF".
class MeshTags:
    """dolfinx.mesh.MeshTags stand-in: tagged entities of one dimension, each given by its vertices."""

    def __init__(self, dim: int, entities: np.ndarray, values: np.ndarray, name: str = "Grid"):
        self.dim = int(dim)
        self.entities = np.ascontiguousarray(entities, dtype=np.int32).reshape(len(values), -1)
        self.values = np.ascontiguousarray(values, dtype=np.int32)
        self.name = name

    def find(self, value: int) -> np.ndarray:
        """Indices (into ``entities``) of the entities tagged ``value`` (MeshTags.find [ext])."""
        return np.nonzero(self.values == int(value))[0]

    def vertices(self, value: int) -> np.ndarray:
        """Sorted unique vertices of the entities tagged ``value``: for CG1 these are the dofs that
        dolfinx.fem.locate_dofs_topological(V, dim, tags.find(value)) returns [ext]."""
        return np.unique(self.entities[self.find(value)]).astype(np.int32)


_CELL_OF = {"triangle": (2, 3), "tetrahedron": (3, 4), "polyline": (1, 2), "line": (1, 2)}


def _read_item(item, directory: str) -> np.ndarray:
    fmt = (item.get("Format") or "XML").upper()
    dims = [int(v) for v in item.get("Dimensions").split()]
    is_int = (item.get("DataType") or item.get("NumberType") or "Float").lower() in ("int", "uint")
    prec = int(item.get("Precision") or (4 if is_int else 8))
    dtype = np.dtype(("<" if (item.get("Endian") or "Little").lower() == "little" else ">") + ("i" if is_int else "f") + str(prec))
    if fmt == "XML":
        return np.array(item.text.split(), dtype=np.float64).astype(dtype).reshape(dims)
    if fmt == "BINARY":
        return np.fromfile(os.path.join(directory, item.text.strip()), dtype=dtype, count=int(np.prod(dims))).reshape(dims)
    raise NotImplementedError(f"XDMF DataItem Format={fmt!r}: HDF5 heavy data cannot be read here (no h5py); convert with "
                              "meshio ... data_format='XML' or write_mesh_files()")


def read_xdmf_grid(path: str):
    """(x, cells, cell type, {attribute name: (centre, values)}) of the first uniform grid of an XDMF file."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    directory = os.path.dirname(os.path.abspath(path))
    grid = next((g for g in root.iter("Grid") if (g.get("GridType") or "Uniform") == "Uniform"), None)
    if grid is None:
        raise ValueError(f"{path}: no uniform grid")
    topo = grid.find("Topology")
    kind = (topo.get("TopologyType") or topo.get("Type")).lower()
    if kind not in _CELL_OF:
        raise NotImplementedError(f"{path}: topology {kind!r} (triangle / tetrahedron / polyline meshes only)")
    cells = _read_item(topo.find("DataItem"), directory).reshape(-1, _CELL_OF[kind][1])
    x = _read_item(grid.find("Geometry").find("DataItem"), directory)
    attrs = {}
    for a in grid.findall("Attribute"):
        attrs[a.get("Name")] = ((a.get("Center") or "Node"), _read_item(a.find("DataItem"), directory).ravel())
    return x, cells, kind, attrs


def read_mesh(path: str):
    """The mesh of an XDMF file (a recorder's ``record_<name>.xdmf`` included) as a ``Mesh``; third coordinates
    that are identically zero are dropped for triangle meshes."""
    from .mesh import Mesh
    x, cells, kind, attrs = read_xdmf_grid(path)
    tdim = _CELL_OF[kind][0]
    if tdim not in (2, 3):
        raise NotImplementedError(f"{path}: a {kind} grid is not a domain mesh")
    if x.shape[1] > tdim:
        if np.any(x[:, tdim:] != 0.0):
            raise NotImplementedError("surface meshes embedded in a higher dimension")
        x = x[:, :tdim]
    return Mesh(np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(cells, dtype=np.int32)), attrs


def _write_grid(path: str, kind: str, x: np.ndarray, cells: np.ndarray, tags, binary: bool) -> None:
    stem = path[:-5]
    base = os.path.basename(stem)
    x3 = np.zeros((x.shape[0], 3))
    x3[:, :x.shape[1]] = x

    def item(arr, dtype, tag):
        dims = " ".join(str(v) for v in arr.shape)
        kindattr = 'DataType="Int" Precision="4"' if dtype == "<i4" else 'DataType="Float" Precision="8"'
        if binary:
            fname = f"{base}_{tag}.bin"
            np.ascontiguousarray(arr, dtype=dtype).tofile(os.path.join(os.path.dirname(path) or ".", fname))
            return f'<DataItem Format="Binary" {kindattr} Endian="Little" Dimensions="{dims}">{fname}</DataItem>'
        body = "\n".join(" ".join(repr(float(v)) if dtype != "<i4" else str(int(v)) for v in np.atleast_1d(row)) for row in arr)
        return f'<DataItem Format="XML" {kindattr} Dimensions="{dims}">\n{body}\n</DataItem>'

    out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', ' <Domain>', '  <Grid Name="Grid" GridType="Uniform">',
           f'   <Topology TopologyType="{kind}" NumberOfElements="{cells.shape[0]}">{item(cells, "<i4", "topology")}</Topology>',
           f'   <Geometry GeometryType="XYZ">{item(x3, "<f8", "geometry")}</Geometry>']
    if tags is not None:
        out.append(f'   <Attribute Name="Grid" AttributeType="Scalar" Center="Cell">{item(np.asarray(tags).reshape(-1, 1), "<i4", "tags")}</Attribute>')
    out += ['  </Grid>', ' </Domain>', '</Xdmf>']
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


def write_mesh_files(prefix: str, mesh, facets: np.ndarray, facet_tags: np.ndarray, association_table: dict,
                     cell_tags=None, directory: str = ".", binary: bool = True) -> None:
    """Writes the three files ``import_mesh`` reads (the layout of the msh2xdmf converter the reference's docstring
    names): the domain, the tagged boundary facets given by their vertices, and the association table."""
    os.makedirs(directory, exist_ok=True)
    dkind = {2: "Triangle", 3: "Tetrahedron"}[mesh.tdim]
    fkind = {2: "Polyline", 3: "Triangle"}[mesh.tdim]
    _write_grid(os.path.join(directory, f"{prefix}_domain.xdmf"), dkind, mesh.x, mesh.conn, cell_tags, binary)
    _write_grid(os.path.join(directory, f"{prefix}_boundaries.xdmf"), fkind, mesh.x, np.asarray(facets, dtype=np.int32), facet_tags, binary)
    with open(os.path.join(directory, f"{prefix}_association_table.ini"), "w") as fh:
        fh.write("[ASSOCIATION TABLE]\n" + "".join(f"{k} = {int(v)}\n" for k, v in association_table.items()))


def import_mesh(prefix="mesh", subdomains=False, dim=2, directory=".", reorder=True):
    """utils_dolfinx.py:69-123 -- same arguments, same return tuple:
    (mesh, boundaries_mf, association_table) or (mesh, boundaries_mf, subdomains_mf, association_table).
    Like dolfinx's reader the mesh comes back renumbered for locality (``reorder``; Mesh.reordered()): the tags
    refer to the new numbering, ``mesh.original_vertex_index`` / ``mesh.original_cell_index`` to the file's."""
    from configparser import ConfigParser
    mesh, attrs = read_mesh(os.path.join(directory, f"{prefix}_domain.xdmf"))
    if mesh.tdim != dim:
        raise ValueError(f"{prefix}_domain.xdmf holds a {mesh.tdim}-D mesh, dim={dim} was asked for")
    _, fcells, fkind, fattrs = read_xdmf_grid(os.path.join(directory, f"{prefix}_boundaries.xdmf"))
    if _CELL_OF[fkind][0] != dim - 1:
        raise ValueError(f"{prefix}_boundaries.xdmf: {fkind} entities are not the facets of a {dim}-D mesh")
    if "Grid" not in fattrs:
        raise ValueError(f"{prefix}_boundaries.xdmf: no 'Grid' tags")
    cell_tags = attrs["Grid"][1] if "Grid" in attrs else None
    if reorder:
        mesh = mesh.reordered()
        fcells = mesh.vertex_perm[fcells]
        if cell_tags is not None:
            cell_tags = cell_tags[mesh.original_cell_index]
    boundaries_mf = MeshTags(dim - 1, fcells, fattrs["Grid"][1])
    file_content = ConfigParser()
    file_content.read(os.path.join(directory, f"{prefix}_association_table.ini"))
    association_table = {k: int(v) for k, v in dict(file_content["ASSOCIATION TABLE"]).items()}
    if not subdomains:
        return mesh, boundaries_mf, association_table
    if cell_tags is None:
        raise ValueError(f"{prefix}_domain.xdmf: no 'Grid' subdomain tags")
    subdomains_mf = MeshTags(dim, mesh.conn, cell_tags)
    return mesh, boundaries_mf, subdomains_mf, association_table
