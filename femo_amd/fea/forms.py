"""Closed catalogue of variational forms (UFL is not available).

A form object plays the role of the UFL form the reference's run scripts build:
it records *which* integrand is meant and the Functions it reads, and
``utils_hip.assemble*`` dispatch on it to the matching HIP kernels.
``derivative(form, function)`` mirrors ``ufl.derivative`` for the Gateaux
derivatives the operators request (utils_dolfinx.py:313-314).

Catalogue
  PoissonResidual        inner(grad u, grad v) dx - inner(f, v) dx
                         (examples/poisson_opt/run_poisson_opt.py:32-38, 62-70)
  L2TrackingFunctional   1/2 (u-u_ex)^2 dx + alpha/2 f^2 dx   (run_poisson_opt.py:74-76)
"""
from __future__ import annotations

from .. import _lib
from .function import Function


class Form:
    rank = None           # 0 scalar, 1 vector, 2 matrix

    def functions(self):
        return ()


class BackendForm(Form):
    """A form that carries its own assembly and solves (the shell forms of fea/shell_forms.py): `utils_hip` hands
    the work over instead of looking the form up in its catalogue.  Interface, by rank --
      rank 0: assemble_scalar() -> float;  assemble_derivative(wrt, out) -> Vec (gradient w.r.t. a Function)
      rank 1: assemble_vector(out) -> Vec;  partial_matrix(wrt, out) -> operator with mult / multTranspose / getSizes;
              assemble_system(bcs, rhs, out, out_nobc) -> (A, b);  solve_state(func, bcs, report);  new_matrix()
    and ``mesh``."""
    mesh = None


class PoissonResidual(Form):
    rank = 1
    pde_kind = _lib.PDE_POISSON
    is_linear = True      # Jacobian independent of u and f
    constant_partials = True   # dR/du, dR/df and A depend on the mesh and the Dirichlet set only (StateOperation: early linearisation)
    is_symmetric = True

    def __init__(self, u: Function, f: Function):
        if u.function_space.family != "CG" or f.function_space.family != "DG":
            raise NotImplementedError("PoissonResidual needs a CG1 state and a DG0 source")
        self.u, self.f = u, f
        self.params = None

    def functions(self):
        return (self.u, self.f)


class NonlinearPoissonResidual(Form):
    """inner(grad u, grad v) dx + inner(u**3, v) dx - inner(f, v) dx  [+ Nitsche boundary terms]
    (examples/nonlinear_poisson_opt/run_nonlinear_poisson_opt.py:88-125).  ``u_exact`` is a CG1
    Function (the reference passes a UFL expression whose quadrature cannot be reproduced without
    FFCx; its interpolant is used).  As in boundaryResidual (:98-117):
    sym=True -> sgn = +1 and the penalty term is on (symmetric Jacobian, CG);
    sym=False -> sgn = -1, penalty only with overPenalize (non-symmetric Jacobian, BiCGSTAB)."""
    rank = 1
    pde_kind = _lib.PDE_NL_POISSON
    is_linear = False

    def __init__(self, u: Function, f: Function, u_exact: Function = None, weak_bc: bool = False,
                 sym: bool = False, beta_value: float = 1e1, overPenalize: bool = False):
        if u.function_space.family != "CG" or f.function_space.family != "DG":
            raise NotImplementedError("NonlinearPoissonResidual needs a CG1 state and a DG0 source")
        if weak_bc and (u_exact is None or u_exact.function_space.family != "CG"):
            raise ValueError("weak_bc=True needs the boundary data u_exact as a CG1 Function")
        self.u, self.f, self.u_exact = u, f, u_exact
        self.weak_bc, self.sym = weak_bc, sym
        self.beta = float(beta_value) if (sym or overPenalize) else 0.0
        self.is_symmetric = bool(sym) or not weak_bc
        self.params = [self.beta, 1.0 if sym else -1.0]

    def functions(self):
        return (self.u, self.f)


class BeamResidual(Form):
    """inner(div grad v, EI div grad u) dx - f v(L),  EI = E width t^3 / 12
    (examples/beam_thickness_opt/run_thickness_opt_cantilever_beam.py:71-79).  ``load`` is the nodal
    load vector of the point force (a Hermite-space Function)."""
    rank = 1
    pde_kind = _lib.PDE_EB_BEAM
    is_linear = True
    is_symmetric = True

    def __init__(self, u: Function, t: Function, load: Function, E: float, width: float):
        if u.function_space.family != "HERMITE" or t.function_space.family != "DG":
            raise NotImplementedError("BeamResidual needs a Hermite-3 state and a DG0 thickness")
        self.u, self.f, self.load = u, t, load
        self.E, self.width = float(E), float(width)
        self.params = [self.E, self.width]

    def functions(self):
        return (self.u, self.f)


class LinearFunctional(Form):
    """J = sum_i coeff_i * arg_i for a fixed coefficient Function of the same space as ``arg``:
    compliance f u(L) (coeff = nodal load) and volume t width L dx (coeff = width * h_e) of the beam
    example (:81-85)."""
    rank = 0

    def __init__(self, coeff: Function, arg: Function, others=()):
        if coeff.function_space.dim != arg.function_space.dim:
            raise ValueError("coefficient and argument live in different spaces")
        self.coeff, self.arg, self.others = coeff, arg, tuple(others)

    def functions(self):
        return (self.arg,) + self.others


class L2TrackingFunctional(Form):
    rank = 0
    functional_kind = _lib.J_L2_TRACKING

    def __init__(self, u: Function, f: Function, u_exact: Function, alpha: float):
        self.u, self.f, self.u_exact, self.alpha = u, f, u_exact, float(alpha)
        self.params = [self.alpha]

    def functions(self):
        return (self.u, self.f, self.u_exact)


class FieldExpression(Form):
    """A field to be L2-projected onto CG1 (fea_dolfinx.py:148-161, utils_dolfinx.py:549-583).
    The reference accepts any UFL expression; the catalogue offers
      FunctionExpr(fn)          a DG0 or CG1 Function itself
      GradientMagnitude(u)      sqrt(inner(grad u, grad u)), cell-wise constant for CG1 u
      PowerExpr(fn, p)          fn ** p for a DG0 Function (run_topo_opt_cantilever_beam.py:257)"""
    rank = 1


class FunctionExpr(FieldExpression):
    def __init__(self, fn: Function):
        self.fn = fn

    def functions(self):
        return (self.fn,)


class GradientMagnitude(FieldExpression):
    def __init__(self, u: Function):
        if u.function_space.family != "CG":
            raise NotImplementedError("GradientMagnitude needs a CG1 Function")
        self.fn = u

    def functions(self):
        return (self.fn,)


class PowerExpr(FieldExpression):
    def __init__(self, fn: Function, p: float):
        if fn.function_space.family != "DG":
            raise NotImplementedError("PowerExpr needs a DG0 Function")
        self.fn, self.p = fn, float(p)

    def functions(self):
        return (self.fn,)


class DerivativeForm(Form):
    """Gateaux derivative of ``form`` w.r.t. ``wrt`` (ufl.derivative)."""

    def __init__(self, form: Form, wrt: Function):
        if not isinstance(wrt, Function) and not hasattr(wrt, "function_space"):
            raise ValueError("derivative is taken w.r.t. a Function")
        # like ufl.derivative, differentiating w.r.t. a Function the form does not depend on is legal
        # (the result assembles to zero); ``depends`` lets the assemblers short-cut it
        self.form, self.wrt = form, wrt
        self.depends = any(wrt is fn for fn in form.functions())
        self.rank = form.rank + 1


def derivative(form: Form, function: Function) -> DerivativeForm:
    return DerivativeForm(form, function)


# --- the user-level builders of examples/poisson_opt -------------------------
ALPHA = 1e-6  # run_poisson_opt.py:28,112


def interiorResidual(u, v, f):
    """run_poisson_opt.py:32-38.  ``v`` (the test function) is implied by the catalogue."""
    return PoissonResidual(u, f)


def pdeRes(u, v, f, u_exact=None, weak_bc=False, sym=False):
    """run_poisson_opt.py:62-70."""
    if weak_bc:
        raise NotImplementedError("Nitsche boundary terms are not in the catalogue yet")
    return interiorResidual(u, v, f)


def outputForm(u, f, u_exact, alpha: float = ALPHA):
    """run_poisson_opt.py:74-76."""
    return L2TrackingFunctional(u, f, u_exact, alpha)


def TestFunction(function_space):
    """Placeholder so run scripts keep their shape; the catalogue fixes the test space."""
    return None
