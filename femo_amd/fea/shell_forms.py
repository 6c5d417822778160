"""Shell forms for the FEA / FEAModel operator stack: the objects `examples/test_shell_m3l/shell_pde.py:219-332`
builds (``ShellPDE`` with its spaces W / VT / VF, ``pdeRes``, ``compliance``, ``mass``, ``volume``,
``elastic_energy``) on the HIP shell kernels (`csrc/shell.hip`, host side `fea/shell.py`).

They are ``BackendForm``s: `utils_hip.assemble*`, `solveNonlinear`, `KSP` hand them their own assembly and solves,
so `FEA.add_input / add_state / add_output`, `StateOperation`, `OutputOperation` and `FEAModel` are used unchanged
(`shell_module.py:20-120` registers exactly these: thickness and nodal force as inputs, ``disp_solid`` as state,
compliance / mass / elastic energy as outputs).

Stress outputs (`shell_pde.py:297-332`): ``pnorm_stress`` (scalar, with partials) and ``von_Mises_stress`` /
``projected_von_Mises_stress`` (field).  Round 3: the boundary conditions as the reference's driver configures them --
``pdeRes(..., penalty=True, dss, dSS, g)`` with ``ShellMeasure`` objects standing for the tagged `ds` / `dS` / `dx`
measures (`shell_pde.py:34,59-61`, `run_pav_shell.py:128-131`) --, ``kinetic_residual`` / ``elastic_residual``
(`shell_pde.py:255-260`), the regularisation options of ``compliance`` (`:262-285`) and its `dxx` subset.  Not built (say so
when asked): the forward-mode product with dR/dh (the reference's forward mode returns zeros, `fea_dolfinx.py:192-206`).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from ..engine import Vec, host_wait
from .forms import BackendForm
from .function import Function
from .shell import DeviceShell, ShellSpace


class ShellMesh:
    """Triangulated surface in R^3 (dolfinx mesh of triangles with gdim 3, `shell_pde.py:220`)."""
    tdim, gdim = 2, 3

    def __init__(self, x, conn):
        self.space = ShellSpace(x, conn)
        self.x, self.conn = self.space.x, self.space.conn
        self.n_vert, self.n_cell = self.space.n_vert, self.space.n_cell
        self._dev = {}

    @classmethod
    def read(cls, path: str) -> "ShellMesh":
        """Triangle grid with three coordinates per vertex from an XDMF file -- what the shell drivers do with
        ``XDMFFile(...).read_mesh(name="Grid")`` (`run_shape_opt_roof.py:40-42`, `shell_pde.py` callers).  Heavy data as
        XML or raw binary (fea/io.py); HDF5 is refused there with a message."""
        from .io import read_xdmf_grid
        x, cells, kind, _ = read_xdmf_grid(path)
        if kind != "triangle":
            raise NotImplementedError(f"{path}: shell meshes are triangle grids (got {kind!r})")
        x = np.asarray(x, dtype=np.float64)
        if x.shape[1] == 2:
            x = np.concatenate([x, np.zeros((x.shape[0], 1))], axis=1)
        return cls(x, cells)

    def write(self, path: str, binary: bool = False) -> None:
        """XDMF3 file of the surface mesh (same writer as the volume meshes')."""
        from .io import _write_grid
        _write_grid(path, "Triangle", self.x, self.conn, None, binary)

    def device(self, ctx) -> DeviceShell:
        d = self._dev.get(id(ctx))
        if d is None:
            d = self._dev[id(ctx)] = DeviceShell(ctx, self.space)
        return d

    def field_space(self) -> "ShellFunctionSpace":
        """The CG1 space field outputs are projected onto (FEA.add_field_output)."""
        return ShellFunctionSpace(self, "VT")

    def centroids(self):
        return self.x[self.conn].mean(axis=1)


class ShellFunctionSpace:
    """kind 'W': CG2^3 x CG1^3 (the state), 'VT': CG1 (thickness), 'VF': CG1^3 (nodal force) -- shell_pde.py:228-230."""

    def __init__(self, mesh: ShellMesh, kind: str):
        if kind not in ("W", "VT", "VF"):
            raise ValueError(f"unknown shell space {kind!r}")
        self.mesh, self.kind = mesh, kind
        self.family = {"W": "SHELL", "VT": "CG", "VF": "CGV"}[kind]
        self.degree = 2 if kind == "W" else 1
        self.num_sub_spaces = {"W": 2, "VT": 0, "VF": 3}[kind]

    @property
    def dim(self) -> int:
        S = self.mesh.space
        return {"W": S.n_dof, "VT": S.n_vert, "VF": 3 * S.n_vert}[self.kind]

    def tabulate_dof_coordinates(self):
        S = self.mesh.space
        if self.kind == "VT":
            return S.x
        if self.kind == "VF":
            return np.repeat(S.x, 3, axis=0)
        return np.concatenate([np.repeat(S.unode_x, 3, axis=0), np.repeat(S.x, 3, axis=0)])

    def __eq__(self, other):
        return isinstance(other, ShellFunctionSpace) and other.mesh is self.mesh and other.kind == self.kind

    __hash__ = object.__hash__


def locate_shell_dofs(W: ShellFunctionSpace, field: str, comp: int, marker) -> np.ndarray:
    """State dofs of one component (field 'u' or 'theta', comp 0..2) at the nodes where ``marker(x)`` holds
    (x: (3, n_nodes), like dolfinx.fem.locate_dofs_geometrical on W.sub(i).sub(comp), run_shape_opt_roof.py:131-146)."""
    S = W.mesh.space
    pts = S.unode_x if field == "u" else S.x
    hit = np.nonzero(np.asarray(marker(pts.T), dtype=bool))[0]
    return (S.u_dof(hit, comp) if field == "u" else S.theta_dof(hit, comp)).astype(np.int32)


def _ctx():
    from .utils_hip import get_context
    return get_context()


def _fixed_mask(space: ShellSpace, bcs) -> Optional[np.ndarray]:
    if not bcs:
        return None
    mask = np.zeros(space.n_dof, dtype=np.uint8)
    for bc in bcs:
        mask[bc.dofs] = 1
    return mask


def _fixed_values(space: ShellSpace, bcs) -> Optional[np.ndarray]:
    vals = np.zeros(space.n_dof)
    any_nonzero = False
    for bc in reversed(list(bcs)):                     # first bc wins on duplicates
        v = np.asarray(bc.values(), dtype=np.float64)
        vals[bc.dofs] = v
        any_nonzero = any_nonzero or bool(np.any(v != 0.0))
    return vals if any_nonzero else None


class ShellMeasure:
    """A UFL measure with subdomain data, as far as the shell forms use it: ``kind`` 'ds' (exterior facets = boundary
    edges), 'dS' (interior facets) or 'dx' (cells), and the entities carrying ``tag``.  ``m(tag)`` selects, like
    ``ds_1(100)`` in `run_aeroelasticity_static_wo_feedback.py:180` / `run_pav_shell.py:144-146`."""

    def __init__(self, mesh: ShellMesh, kind: str, entities, tag: int = 0, selected: bool = False):
        if kind not in ("ds", "dS", "dx"):
            raise ValueError(f"unknown measure {kind!r}")
        self.mesh, self.kind, self.tag, self.selected = mesh, kind, int(tag), selected
        self.entities = np.unique(np.asarray(entities, dtype=np.int64))

    def __call__(self, tag: int) -> "ShellMeasure":
        ents = self.entities if int(tag) == self.tag else np.zeros(0, np.int64)
        return ShellMeasure(self.mesh, self.kind, ents, tag, selected=True)

    def cell_weight(self) -> np.ndarray:
        """DG0 indicator of a 'dx' measure."""
        if self.kind != "dx":
            raise ValueError("cell_weight: not a dx measure")
        w = np.zeros(self.mesh.n_cell)
        w[self.entities] = 1.0
        return w


def createCustomMeasure(mesh: ShellMesh, dim: int, marker, measure: str = "ds", tag: int = 100) -> ShellMeasure:
    """`createCustomMeasure(mesh, fdim, locator, measure='ds' | 'dS' | 'dx', tag=...)` of the shell drivers
    (`run_pav_shell.py:128-131`): the entities of dimension ``dim`` all of whose vertices satisfy ``marker(x)`` (x: (3, n));
    'ds' keeps the boundary edges among them, 'dS' the interior ones (locate_entities_boundary / locate_entities [ext],
    `run_aeroelasticity_static_wo_feedback.py:110-124`)."""
    S = mesh.space
    if measure in ("ds", "dS"):
        if dim != 1:
            raise ValueError("facets of a surface mesh have dimension 1")
        ext, inte = S.tagged_edges(marker)
        return ShellMeasure(mesh, measure, ext if measure == "ds" else inte, tag)
    if measure == "dx":
        if dim != 2:
            raise ValueError("cells of a surface mesh have dimension 2")
        hit = np.asarray(marker(S.x.T), dtype=bool)
        return ShellMeasure(mesh, "dx", np.nonzero(hit[S.conn].all(axis=1))[0], tag)
    raise ValueError(f"unknown measure {measure!r}")


PENALTY_BETA = 1e15        # the value the reference's recorded penalty runs name (run_aeroelasticity_static_wo_feedback.py:466,499)


class ShellMatrix:
    """dR/dw = K(h) on the element-coupling pattern; with ``fixed`` the strongly imposed dofs are identity rows /
    columns in every product and solve (the A of state_model.py:149)."""
    symmetric = True
    pde_kind = None

    def __init__(self, form: "ShellResidual", fixed: Optional[np.ndarray] = None, xfix: Optional[np.ndarray] = None):
        self.form, self.fixed, self.xfix = form, fixed, xfix
        self.mesh = form.mesh
        self._row = self._col = None
        self.info = None

    def getSizes(self):
        n = self.mesh.space.n_dof
        return (n, n)

    size = property(getSizes)

    def _masked(self, x: Vec) -> Vec:
        return x

    def mult(self, x: Vec, y: Vec) -> Vec:
        dev = self.mesh.device(_ctx())
        if self.fixed is None:
            return dev.matvec(self.form.stiffness(), x, y)
        # identity rows / columns: y = M K M x + (I - M) x through host masks (products with A are not on the hot path)
        xm = np.array(x.get()); free = self.fixed == 0
        tmp = Vec(_ctx(), xm.size).set(xm * free)
        dev.matvec(self.form.stiffness(), tmp, y)
        yh = np.array(y.get())
        y.set(np.where(free, yh, xm))
        return y

    multTranspose = mult

    def new_row_vec(self) -> Vec:
        if self._row is None:
            self._row = Vec(_ctx(), self.mesh.space.n_dof)
        return self._row

    def new_col_vec(self) -> Vec:
        if self._col is None:
            self._col = Vec(_ctx(), self.mesh.space.n_dof)
        return self._col

    def backend_solve(self, b: Vec, x: Vec, options: Optional[dict] = None) -> None:
        dev = self.mesh.device(_ctx())
        o = options or {}
        xfix = Vec(_ctx(), self.mesh.space.n_dof).set(self.xfix) if self.xfix is not None else None
        self.info = dev.solve(self.form.stiffness(), b, x, fixed=self.fixed, xfix=xfix,
                              rtol=o.get("shell_rtol", 1e-12), max_it=o.get("shell_max_it", 2_000_000))

    def to_scipy(self):
        import scipy.sparse as sp
        rowptr, cols, _ = self.mesh.space.pattern()
        n = self.mesh.space.n_dof
        return sp.csr_matrix((np.array(self.form.stiffness().get()), cols, rowptr), shape=(n, n))


class _ShellPartial:
    """dR/dh (n_dof x n_vert) or dR/df (n_dof x 3 n_vert) of the shell residual, matrix free."""

    def __init__(self, form: "ShellResidual", wrt: str):
        self.form, self.wrt, self.mesh = form, wrt, form.mesh
        self._row = self._col = None

    def getSizes(self):
        S = self.mesh.space
        return (S.n_dof, S.n_vert if self.wrt == "h" else 3 * S.n_vert)

    def mult(self, x: Vec, y: Vec) -> Vec:
        dev, F = self.mesh.device(_ctx()), self.form
        if self.wrt == "h":
            # d_residuals += dR/dh . d_h (state_model.py:176-188, fwd mode): (dK/dh [d_h]) w, element by element
            return dev.dform_dh_fwd(F.E, F.nu, F.h.vec, x, F.w.vec, y)
        return dev.load(x, y, sign=-1.0)

    def multTranspose(self, x: Vec, y: Vec) -> Vec:
        dev, F = self.mesh.device(_ctx()), self.form
        if self.wrt == "h":
            return dev.dform_dh(F.E, F.nu, F.h.vec, x, F.w.vec, out=y)
        return dev.load_T(x, y, sign=-1.0)

    def new_row_vec(self) -> Vec:
        if self._row is None:
            self._row = Vec(_ctx(), self.getSizes()[0])
        return self._row

    def new_col_vec(self) -> Vec:
        if self._col is None:
            self._col = Vec(_ctx(), self.getSizes()[1])
        return self._col


class ShellResidual(BackendForm):
    """R(w; h, f) = dE_elastic(w; h)/dw - int f . v = K(h) w - F(f)   (pdeRes -> weakFormResidual, shell_pde.py:246-253)."""
    rank = 1
    is_linear = True
    is_symmetric = True

    def __init__(self, h: Function, w: Function, f: Optional[Function], E: float, nu: float, penalty_edges=None,
                 g: Optional[Function] = None, beta: float = PENALTY_BETA, with_load: bool = True):
        """``penalty_edges``: ids of the edges of the `dss` / `dSS` measures -- the penalty form of the boundary conditions,
        K_pen (w - g) added to the residual and K_pen to dR/dw (oracle/shell_oracle.py::penalty_matrix); ``with_load=False``:
        the elastic force alone (`elastic_residual`, shell_pde.py:258-260)."""
        self.h, self.w, self.f, self.E, self.nu = h, w, f, float(E), float(nu)
        self.u = w
        self.mesh = w.function_space.mesh
        self._vals: Optional[Vec] = None
        self._vals_ver = None
        self._res: Optional[Vec] = None
        self.penalty_edges = None if penalty_edges is None else np.unique(np.asarray(penalty_edges, dtype=np.int64))
        self.g, self.beta, self.with_load = g, float(beta), bool(with_load)

    @property
    def has_penalty(self) -> bool:
        return self.penalty_edges is not None and self.penalty_edges.size > 0

    def functions(self):
        return tuple(fn for fn in (self.h, self.w, self.f, self.g) if fn is not None)

    def _bind_penalty(self, dev: DeviceShell) -> None:
        """The device handle holds one set of tagged edges: (re)load this form's when another form's is there."""
        # keyed on content, not on id(self): CPython reuses addresses, and the handle outlives the forms
        edges = self.penalty_edges if self.has_penalty else np.zeros(0, np.int64)
        key = (self.beta, int(edges.size), hash(edges.tobytes()))
        if getattr(dev, "_penalty_owner", None) != key:
            dev.set_penalty(edges, self.beta)
            dev._penalty_owner = key

    def stiffness(self) -> Vec:
        """dR/dw = K(h) (+ K_pen): reassembled when the thickness changed."""
        dev = self.mesh.device(_ctx())
        if self._vals is None:
            self._vals = Vec(_ctx(), dev.nnz)
        if self._vals_ver != self.h.version:
            dev.assemble(self.E, self.nu, self.h.vec, self._vals)
            if self.has_penalty:
                self._bind_penalty(dev)
                dev.penalty_add(self._vals)
            self._vals_ver = self.h.version
        return self._vals

    def _rhs(self, out: Vec) -> Vec:
        """F(f) + K_pen g: what the state equation K_total w = rhs has on its right."""
        dev = self.mesh.device(_ctx())
        if self.with_load and self.f is not None:
            dev.load(self.f.vec, out)
        else:
            out.fill(0.0)
        if self.has_penalty and self.g is not None:
            self._bind_penalty(dev)
            dev.penalty_apply(self.g.vec, out, accumulate=True)   # + K_pen g
        return out

    def new_matrix(self) -> ShellMatrix:
        return ShellMatrix(self)

    def assemble_vector(self, out: Optional[Vec] = None) -> Vec:
        dev = self.mesh.device(_ctx())
        if out is None:
            if self._res is None:
                self._res = Vec(_ctx(), self.mesh.space.n_dof)
            out = self._res
        dev.matvec(self.stiffness(), self.w.vec, out)              # (K + K_pen) w
        if self.with_load and self.f is not None:
            dev.load(self.f.vec, out, sign=-1.0, accumulate=True)
        if self.has_penalty and self.g is not None:
            self._bind_penalty(dev)
            tmp = Vec(_ctx(), self.mesh.space.n_dof)
            dev.penalty_apply(self.g.vec, tmp)                     # K_pen g
            out.axpy(-1.0, tmp)
        return out

    def partial_matrix(self, wrt: Function, out=None):
        if wrt is self.w:
            return ShellMatrix(self)
        if wrt is self.h:
            return out if isinstance(out, _ShellPartial) and out.wrt == "h" else _ShellPartial(self, "h")
        if self.f is not None and wrt is self.f:
            if not self.with_load:
                raise ValueError("the elastic residual does not depend on the load")
            return out if isinstance(out, _ShellPartial) and out.wrt == "f" else _ShellPartial(self, "f")
        raise ValueError("the shell residual does not depend on that Function")

    def assemble_system(self, bcs, rhs: bool, out, out_nobc):
        if rhs:
            raise NotImplementedError("assembleSystem(rhs=True) for the shell form: use solveNonlinear / FEA.solve")
        self.stiffness()
        S = self.mesh.space
        A = out if isinstance(out, ShellMatrix) else ShellMatrix(self)
        A.form, A.fixed, A.xfix = self, _fixed_mask(S, bcs), None      # linearised solves: homogeneous values
        if isinstance(out_nobc, ShellMatrix):
            out_nobc.form, out_nobc.fixed = self, None
        return A, None

    def solve_state(self, func: Function, bcs, report: bool = False) -> None:
        """K(h) w = F(f) with the imposed dofs (solveNonlinear -> NewtonSolver on a linear residual: one solve)."""
        dev = self.mesh.device(_ctx())
        S = self.mesh.space
        F = self._rhs(Vec(_ctx(), S.n_dof))
        vals = _fixed_values(S, bcs) if bcs else None
        xfix = Vec(_ctx(), S.n_dof).set(vals) if vals is not None else None
        info = dev.solve(self.stiffness(), F, func.vec, fixed=_fixed_mask(S, bcs), xfix=xfix)
        func.version += 1
        if report:
            print(f"shell solve: {info.iterations} CG iterations, {info.solve_ms:.1f} ms")


class _ShellScalar(BackendForm):
    rank = 0

    def assemble_derivative(self, wrt: Function, out: Optional[Vec] = None) -> Vec:
        n = wrt.function_space.dim
        if out is None:
            cache = self.__dict__.setdefault("_grad_vecs", {})
            out = cache.get(id(wrt))
            if out is None or out.n != n:
                out = cache[id(wrt)] = Vec(_ctx(), n)
        if not any(wrt is fn for fn in self.functions()):
            return out.fill(0.0)
        return self._gradient(wrt, out)


class ShellCompliance(_ShellScalar):
    """1/2 int_dxx u_mid . u_mid + regularization(h, type)  (shell_pde.py:262-285): ``dxx`` a tagged 'dx' measure (None: the
    whole surface), ``regularization`` None (the reference's default), 'H1', 'L2H1' or 'L2'."""

    def __init__(self, w: Function, h: Optional[Function] = None, dxx: Optional[ShellMeasure] = None,
                 regularization: Optional[str] = None):
        self.w, self.h, self.mesh = w, h, w.function_space.mesh
        if regularization is not None and regularization not in DeviceShell.REGULARIZATION_KINDS:
            raise ValueError(f"unknown regularisation {regularization!r}")
        if regularization is not None and h is None:
            raise ValueError("the regularisation term needs the thickness")
        self.regularization = regularization
        if dxx is not None and (not isinstance(dxx, ShellMeasure) or dxx.kind != "dx"):
            raise TypeError("compliance: dxx must be a ShellMeasure of kind 'dx'")
        self._cw_host = None if dxx is None else dxx.cell_weight()
        self._cw: Optional[Vec] = None

    def functions(self):
        return (self.w,) if self.regularization is None else (self.w, self.h)

    def _weight(self) -> Optional[Vec]:
        if self._cw_host is None:
            return None
        if self._cw is None:
            self._cw = Vec(_ctx(), self.mesh.n_cell).set(self._cw_host)
        return self._cw

    def assemble_scalar(self) -> float:
        dev = self.mesh.device(_ctx())
        J = dev.compliance_dx(self.w.vec, self._weight())
        if self.regularization is not None:
            J += dev.regularization(self.regularization, self.h.vec)
        return J

    def _gradient(self, wrt, out):
        dev = self.mesh.device(_ctx())
        if wrt is self.w:
            dev.compliance_dx(self.w.vec, self._weight(), grad=out, value=False)
        else:
            dev.regularization(self.regularization, self.h.vec, grad=out, value=False)
        return out


class ShellInertialResidual(BackendForm):
    """`kinetic_residual(rho, h)` (shell_pde.py:255-256 -> inertialResidual): M(h) a for the accelerations ``acc`` (a Function
    on W: uddot on the displacement dofs, thetaddot on the rotations; the reference's dynamic driver passes them
    explicitly, run_aeroelasticity_dynamic.py:93).  Partials: w.r.t. ``acc`` the (symmetric) mass operator, w.r.t. the
    thickness lam^T dM/dh a."""
    rank = 1
    is_linear = True
    is_symmetric = True

    def __init__(self, rho: float, h: Function, acc: Function):
        self.rho, self.h, self.acc = float(rho), h, acc
        self.u = acc
        self.mesh = acc.function_space.mesh
        self._res: Optional[Vec] = None

    def functions(self):
        return (self.h, self.acc)

    def assemble_vector(self, out: Optional[Vec] = None) -> Vec:
        if out is None:
            if self._res is None:
                self._res = Vec(_ctx(), self.mesh.space.n_dof)
            out = self._res
        return self.mesh.device(_ctx()).inertia_apply(self.rho, self.h.vec, self.acc.vec, out)

    def partial_matrix(self, wrt: Function, out=None):
        if wrt is self.acc:
            return _InertiaOperator(self, "a")
        if wrt is self.h:
            return _InertiaOperator(self, "h")
        raise ValueError("the inertial residual does not depend on that Function")


class _InertiaOperator:
    def __init__(self, form: ShellInertialResidual, wrt: str):
        self.form, self.wrt, self.mesh = form, wrt, form.mesh
        self._row = self._col = None

    def getSizes(self):
        S = self.mesh.space
        return (S.n_dof, S.n_dof if self.wrt == "a" else S.n_vert)

    def mult(self, x: Vec, y: Vec) -> Vec:
        F = self.form
        if self.wrt == "h":
            return self.mesh.device(_ctx()).inertia_dh_fwd(F.rho, F.h.vec, x, F.acc.vec, y)
        return self.mesh.device(_ctx()).inertia_apply(F.rho, F.h.vec, x, y)

    def multTranspose(self, x: Vec, y: Vec) -> Vec:
        F = self.form
        dev = self.mesh.device(_ctx())
        if self.wrt == "a":
            return dev.inertia_apply(F.rho, F.h.vec, x, y)
        return dev.inertia_dh(F.rho, F.h.vec, x, F.acc.vec, y)

    def new_row_vec(self) -> Vec:
        if self._row is None:
            self._row = Vec(_ctx(), self.getSizes()[0])
        return self._row

    def new_col_vec(self) -> Vec:
        if self._col is None:
            self._col = Vec(_ctx(), self.getSizes()[1])
        return self._col


class ShellRegularization(_ShellScalar):
    """`ShellPDE.regularization(h, type)` (shell_pde.py:262-282) as a scalar form: 'H1', 'L2H1' or 'L2'."""

    def __init__(self, h: Function, kind: str):
        if kind not in DeviceShell.REGULARIZATION_KINDS:
            raise ValueError(f"unknown regularisation {kind!r}")
        self.h, self.kind, self.mesh = h, kind, h.function_space.mesh

    def functions(self):
        return (self.h,)

    def assemble_scalar(self) -> float:
        return self.mesh.device(_ctx()).regularization(self.kind, self.h.vec)

    def _gradient(self, wrt, out):
        self.mesh.device(_ctx()).regularization(self.kind, self.h.vec, grad=out, value=False)
        return out


class ShellMass(_ShellScalar):
    """int rho h dx (shell_pde.py:293-294); rho = 1: the volume (:290-291)."""

    def __init__(self, h: Function, rho: float = 1.0):
        self.h, self.rho, self.mesh = h, float(rho), h.function_space.mesh

    def functions(self):
        return (self.h,)

    def assemble_scalar(self) -> float:
        return self.mesh.device(_ctx()).mass(self.rho, self.h.vec)

    def _gradient(self, wrt, out):
        self.mesh.device(_ctx()).mass(self.rho, self.h.vec, grad=out, value=False)
        return out


class ShellElasticEnergy(_ShellScalar):
    """1/2 w^T K(h) w (shell_pde.py:296-299)."""

    def __init__(self, w: Function, h: Function, E: float, nu: float, residual: Optional["ShellResidual"] = None):
        self.w, self.h, self.E, self.nu, self.mesh = w, h, float(E), float(nu), w.function_space.mesh
        # dE/dw = K(h) w needs the stiffness WITHOUT penalty terms: share the residual form's value array when it has
        # none (a second nnz-sized copy, ~1 GB at 2 M dofs), else keep a form of its own
        share = residual is not None and not residual.has_penalty and residual.h is h and residual.E == self.E and residual.nu == self.nu
        self._res = residual if share else ShellResidual(h, w, None, E, nu)

    def functions(self):
        return (self.w, self.h)

    def assemble_scalar(self) -> float:
        return self.mesh.device(_ctx()).dform_dh(self.E, self.nu, self.h.vec, self.w.vec, self.w.vec, energy=True)

    def _gradient(self, wrt, out):
        dev = self.mesh.device(_ctx())
        if wrt is self.w:
            return dev.matvec(self._res.stiffness(), self.w.vec, out)
        dev.dform_dh(self.E, self.nu, self.h.vec, self.w.vec, self.w.vec, out=out)          # w^T dK/dh w
        half = Vec(_ctx(), out.n).copy_from(out)
        return out.axpy(-0.5, half)                                                         # ... times 1/2


class ShellPnormStress(_ShellScalar):
    """1 / alpha int (m sigma_vm)^rho dx (shell_pde.py:297-313): the aggregated von Mises stress on the top (surface = +1),
    mid (0) or bottom (-1) surface; alpha = surface area unless given."""

    def __init__(self, w: Function, h: Function, E: float, nu: float, m: float = 1e-6, rho: float = 100.0,
                 alpha: Optional[float] = None, surface: float = 1.0, regularization: bool = False):
        self.w, self.h, self.E, self.nu, self.mesh = w, h, float(E), float(nu), w.function_space.mesh
        self.m, self.rho, self.surface = float(m), float(rho), float(surface)
        self.regularization = bool(regularization)          # + 1/alpha int 0.5 * 1e3 * h**rho dx (shell_pde.py:307-309)
        if alpha is None:
            x, c = self.mesh.x, self.mesh.conn
            alpha = float(0.5 * np.linalg.norm(np.cross(x[c[:, 1]] - x[c[:, 0]], x[c[:, 2]] - x[c[:, 0]]), axis=1).sum())
        self.alpha = float(alpha)

    def functions(self):
        return (self.w, self.h)

    def assemble_scalar(self) -> float:
        dev = self.mesh.device(_ctx())
        J = dev.pnorm_stress(self.E, self.nu, self.h.vec, self.w.vec, self.m, self.rho, self.alpha, self.surface)
        if self.regularization:
            J += dev.hpower(0.5e3 / self.alpha, self.rho, self.h.vec)
        return J

    def _gradient(self, wrt, out):
        dev = self.mesh.device(_ctx())
        kw = dict(grad_w=out) if wrt is self.w else dict(grad_h=out)
        dev.pnorm_stress(self.E, self.nu, self.h.vec, self.w.vec, self.m, self.rho, self.alpha, self.surface, value=False, **kw)
        if self.regularization and wrt is self.h:
            dev.hpower(0.5e3 / self.alpha, self.rho, self.h.vec, grad=out, value=False, accumulate=True)
        return out


class ShellVonMises(BackendForm):
    """The von Mises stress on the top / mid / bottom surface as a field expression (shell_pde.py:315-328); `project`
    (and with it `FEA.add_field_output`, fea_dolfinx.py:148-161) hands the L2 projection onto CG1 over to it."""
    rank = 0

    def __init__(self, w: Function, h: Function, E: float, nu: float, surface: float):
        self.w, self.h, self.E, self.nu, self.surface = w, h, float(E), float(nu), float(surface)
        self.mesh = w.function_space.mesh

    def functions(self):
        return (self.w, self.h)

    def project_field(self, target: Function, lump_mass: bool = False) -> Function:
        if target.function_space.dim != self.mesh.space.n_vert:
            raise NotImplementedError("the von Mises stress is projected onto the CG1 space of the shell mesh (pde.VT)")
        self.mesh.device(_ctx()).project_von_mises(self.E, self.nu, self.h.vec, self.w.vec, self.surface, target.vec, lump_mass=lump_mass)
        return target


class ShellPDE:
    """`shell_pde.py:219-332` on the HIP engine: spaces and form builders with the reference's names."""

    def __init__(self, mesh: ShellMesh):
        self.mesh = mesh
        self.W = ShellFunctionSpace(mesh, "W")
        self.VT = ShellFunctionSpace(mesh, "VT")
        self.VF = ShellFunctionSpace(mesh, "VF")

    def _cell_areas(self) -> np.ndarray:
        x, c = self.mesh.x, self.mesh.conn
        return 0.5 * np.linalg.norm(np.cross(x[c[:, 1]] - x[c[:, 0]], x[c[:, 2]] - x[c[:, 0]]), axis=1)

    @property
    def bf_sup_sizes(self) -> np.ndarray:
        """shell_pde.py:233-234: int phi_i dx for the CG1 basis -- a third of the area of the triangles around vertex i
        (the support sizes the drivers use to turn nodal forces into tractions)."""
        if getattr(self, "_bf_sup", None) is None:
            out = np.zeros(self.mesh.n_vert)
            np.add.at(out, self.mesh.conn.ravel(), np.repeat(self._cell_areas() / 3.0, 3))
            self._bf_sup = out
        return self._bf_sup

    def compute_alpha(self) -> float:
        """shell_pde.py:237-244: CellDiameter projected onto CG1 with the consistent mass matrix (`project(..., lump_mass=
        False)`), alpha = mean(nodal values)^2 / 2 -- the drivers' cell-area based parameter.  Host arithmetic on the surface
        mesh (P1 mass matrix, conjugate gradients): called once per model, not on the hot path."""
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        x, c = self.mesh.x, self.mesh.conn
        area = self._cell_areas()
        e = [np.linalg.norm(x[c[:, i]] - x[c[:, j]], axis=1) for i, j in ((0, 1), (1, 2), (2, 0))]
        diam = np.maximum(np.maximum(e[0], e[1]), e[2])                         # CellDiameter of a triangle: its longest edge [ext UFL]
        nv = self.mesh.n_vert
        rows = np.repeat(c, 3, axis=1).ravel()
        cols = np.tile(c, (1, 3)).ravel()
        loc = (np.ones((3, 3)) + np.eye(3)) / 12.0                             # P1 element mass matrix / area
        M = sp.csr_matrix(((area[:, None, None] * loc[None]).ravel(), (rows, cols)), shape=(nv, nv))
        b = np.zeros(nv)
        np.add.at(b, c.ravel(), np.repeat(diam * area / 3.0, 3))
        hn, info = spla.cg(M, b, rtol=1e-12, atol=0.0, M=sp.diags(1.0 / M.diagonal()))
        if info != 0:
            raise RuntimeError("compute_alpha: the projection did not converge")
        return float(np.average(hn) ** 2 / 2.0)

    def compute_nodal_disp(self, func) -> List[np.ndarray]:
        """shell_pde.py:333-334 (`computeNodalDisp` of the absent shell_analysis_fenicsx, used as
        ``uZ = computeNodalDisp(state.sub(0))[2]``, run_shape_opt_roof.py:222): the three displacement components at the
        mesh vertices.  ``func``: the state Function on W (its displacement part is what is returned)."""
        w = np.asarray(host_wait(func.vec.get())) if hasattr(func, "vec") else np.asarray(func, dtype=np.float64)
        u = self.mesh.space.vertex_displacement(w)
        return [np.ascontiguousarray(u[:, k]) for k in range(3)]

    def pdeRes(self, h, w, f, E, nu, penalty=False, dss=None, dSS=None, g=None, beta: float = PENALTY_BETA) -> ShellResidual:
        """shell_pde.py:246-253.  ``penalty=True``: the boundary conditions w = g as penalty terms on the tagged exterior
        (``dss``) and interior (``dSS``) facets, as the reference's drivers configure the shell (shell_pde.py:34,59-61);
        the measures are ``ShellMeasure`` objects (``createCustomMeasure``); untagged `ufl.ds` / `ufl.dS` defaults of the
        reference correspond to passing every boundary / interior edge.  ``beta``: the penalty parameter (the form itself
        is in the absent shell_analysis_fenicsx; the recorded runs name 1e15)."""
        edges = None
        if penalty:
            parts = []
            for m, kind in ((dss, "ds"), (dSS, "dS")):
                if m is None:
                    continue
                if not isinstance(m, ShellMeasure) or m.kind != kind:
                    raise TypeError(f"pdeRes: {kind} must be a ShellMeasure of kind {kind!r}")
                parts.append(m.entities)
            if not parts:
                raise ValueError("pdeRes(penalty=True) needs the tagged measures dss and / or dSS")
            edges = np.concatenate(parts)
        self.elastic_model = res = ShellResidual(h, w, f, E, nu, penalty_edges=edges, g=g, beta=beta)
        return res

    def kinetic_residual(self, rho, h, acc: Optional[Function] = None) -> "ShellInertialResidual":
        """shell_pde.py:255-256.  ``acc``: the accelerations (uddot, thetaddot) as a Function on W; a new zero Function by
        default (the reference's method takes them from the elastic model it built in pdeRes)."""
        return ShellInertialResidual(rho, h, acc if acc is not None else Function(self.W))

    def elastic_residual(self, h, w, f, E, nu, penalty=False, dss=None, dSS=None, g=None) -> ShellResidual:
        """shell_pde.py:258-260: pdeRes without penalty terms plus int f . du_mid, i.e. the elastic force K(h) w alone."""
        return ShellResidual(h, w, f, E, nu, with_load=False)

    def regularization(self, h, type=None):
        """shell_pde.py:262-282 as a scalar form of its own (0.0 for ``type=None``, like the reference)."""
        if type is None:
            return 0.0
        return ShellRegularization(h, type)

    def compliance(self, u_mid, h=None, dxx=None, regularization=None) -> ShellCompliance:
        """shell_pde.py:284-285: 1/2 int_dxx u_mid . u_mid + regularization(h) (the reference's default type is None).
        ``u_mid``: the state Function (its displacement part is what enters, like `ufl.split(w)[0]`)."""
        return ShellCompliance(u_mid, h, dxx, regularization)

    def volume(self, h) -> ShellMass:
        return ShellMass(h, 1.0)

    def mass(self, h, rho) -> ShellMass:
        return ShellMass(h, rho)

    def elastic_energy(self, w, h, E, nu=None) -> ShellElasticEnergy:
        """shell_pde.py:296-299: the elastic model of the last ``pdeRes`` supplies the Poisson ratio, as in the reference
        (where ``elastic_energy(w, h, E)`` reuses ``self.elastic_model``); without one nu defaults to 0."""
        em = getattr(self, "elastic_model", None)
        if nu is None:
            nu = em.nu if em is not None else 0.0
        return ShellElasticEnergy(w, h, E, nu, residual=em if em is not None and em.w is w else None)

    def pnorm_stress(self, w, h, E, nu, dx=None, m=1e-6, rho=100, alpha=None, regularization=False, surface='Top') -> ShellPnormStress:
        """shell_pde.py:297-313 (stress on the top surface there; 'Mid' / 'Bot' as in von_Mises_stress, :315-328)."""
        return ShellPnormStress(w, h, E, nu, m=m, rho=rho, alpha=alpha, surface={'Top': 1.0, 'Mid': 0.0, 'Bot': -1.0}[surface],
                                regularization=bool(regularization))

    def von_Mises_stress(self, w, h, E, nu, surface='Top') -> ShellVonMises:
        """shell_pde.py:315-328"""
        if surface not in ('Top', 'Mid', 'Bot'):
            raise TypeError("Unsupported surface type for stress computation.")
        return ShellVonMises(w, h, E, nu, {'Top': 1.0, 'Mid': 0.0, 'Bot': -1.0}[surface])

    def projected_von_Mises_stress(self, vm_stress: ShellVonMises) -> Function:
        """shell_pde.py:330-332: the consistent L2 projection onto VT."""
        return vm_stress.project_field(Function(self.VT), lump_mass=False)
