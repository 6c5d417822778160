"""Function spaces and functions (replaces dolfinx.fem.FunctionSpace / Function).

Only the two spaces on the hot path exist: ("CG", 1) on vertices and ("DG", 0)
on cells (run_poisson_opt.py:98-105).  A Function owns one device vector; its
``vector`` attribute offers the three PETSc idioms the reference uses
(utils_dolfinx.py:155-167, 308): ``getArray()``, ``[:] = array`` and ``set(x)``.
"""
from __future__ import annotations

import numpy as np

from ..engine import Vec


class FunctionSpace:
    def __init__(self, mesh, element=("CG", 1)):
        family, degree = element
        from .mesh import BeamMesh
        if (family, degree) == ("Hermite", 3):
            if not isinstance(mesh, BeamMesh):
                raise NotImplementedError("the cubic Hermite element is implemented on interval (beam) meshes only")
        elif (family, degree) not in (("CG", 1), ("Lagrange", 1), ("DG", 0)):
            raise NotImplementedError(
                f"function space {element}: the HIP engine implements CG1 / Hermite-3 states and DG0 inputs only")
        elif isinstance(mesh, BeamMesh) and family != "DG":
            raise NotImplementedError("on a beam mesh the state space is ('Hermite', 3)")
        self.mesh = mesh
        self.family = "DG" if family == "DG" else ("HERMITE" if family == "Hermite" else "CG")
        self.degree = degree
        self.num_sub_spaces = 0

    @property
    def dim(self) -> int:
        return self.mesh.n_cell if self.family == "DG" else self.mesh.n_vert

    def tabulate_dof_coordinates(self) -> np.ndarray:
        if self.family == "DG":
            return self.mesh.centroids()
        return self.mesh.x[:, :1] if self.family == "HERMITE" else self.mesh.x

    def __eq__(self, other):
        return (isinstance(other, FunctionSpace) and other.mesh is self.mesh
                and other.family == self.family and other.degree == self.degree)

    __hash__ = object.__hash__


class _VectorView:
    """PETSc-Vec-flavoured access to a Function's device vector."""

    def __init__(self, fn: "Function"):
        self._fn = fn

    def getArray(self) -> np.ndarray:
        # the reference refreshes ghosts before handing the array out (utils_dolfinx.py:155-159 -> 200, 205: ghostUpdate);
        # here it also means the host copy and the device vector agree INCLUDING the ghost tail, at one generation: later
        # operators find the ghosts fresh (no further exchange) and the host copy a valid mirror (no re-upload)
        self.ghostUpdate()
        return self._fn.vec.get()

    def set(self, value) -> None:
        """PETSc ``Vec.set(alpha)`` [ext]: one scalar for every entry (a length-1 array is accepted, as
        ``update`` passes one, utils_dolfinx.py:308-309); arrays go through ``setFuncArray`` / ``v[:] =``."""
        a = np.asarray(value)
        if a.size != 1:
            raise ValueError(f"Vec.set takes a scalar, got an array of size {a.size}; use setFuncArray")
        self._fn.version += 1
        self._fn.vec.fill(float(a.ravel()[0]))

    def __setitem__(self, key, value) -> None:
        if key != slice(None):
            raise NotImplementedError("only v.vector[:] = array is supported")
        arr = np.asarray(value, dtype=np.float64)
        self._fn.version += 1
        if arr.ndim == 0:
            self._fn.vec.fill(float(arr))
        else:
            self._fn.vec.set(arr)

    # no-ops kept so reference call sequences read the same (utils_dolfinx.py:166-167)
    def assemble(self) -> None:
        pass

    def ghostUpdate(self, *a, **k) -> None:
        """Ghost entries <- the owners' values on a partitioned mesh (PETSc ``Vec.ghostUpdate`` [ext]); a no-op on one rank, for
        cell-wise (DG0) functions, and when nothing has written the vector since its last refresh (femo_halo_exchange)."""
        fs = self._fn.function_space
        mesh = fs.mesh
        local = getattr(mesh, "local", None)
        dm = getattr(mesh, "_device", None)
        if local is None or local.nranks <= 1 or dm is None or fs.family != "CG" or len(local.nbr) == 0:
            return
        if self._fn.vec.n >= dm.n_vert:
            dm.halo_exchange(self._fn.vec)


class _XView:
    def __init__(self, fn):
        self._fn = fn

    @property
    def array(self):
        return _ArrayProxy(self._fn)


class _ArrayProxy:
    def __init__(self, fn):
        self._fn = fn

    def __setitem__(self, key, value):
        self._fn.vector[key] = value

    def __array__(self, dtype=None, copy=None):
        return self._fn.vector.getArray()


class Function:
    def __init__(self, function_space: FunctionSpace, name: str = "f"):
        from .utils_hip import get_context
        self.function_space = function_space
        self.name = name
        self.vec = Vec(get_context(), function_space.dim)
        self.version = 0          # bumped by every host-side write (cache key for Dirichlet values)
        self.vector = _VectorView(self)
        self.x = _XView(self)

    def interpolate(self, fn) -> None:
        """fn receives coordinates shaped (3, n_dofs) like dolfinx [ext] (fea_dolfinx.py:163-167)."""
        x = self.function_space.tabulate_dof_coordinates()
        xt = np.zeros((3, x.shape[0]))
        xt[:x.shape[1]] = x.T
        self.version += 1
        self.vec.set(np.asarray(fn(xt), dtype=np.float64))

    def rename(self, name, label=None) -> None:
        self.name = name
