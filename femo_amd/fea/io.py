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
