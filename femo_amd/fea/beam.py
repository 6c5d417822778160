"""The form builders of examples/beam_thickness_opt/run_thickness_opt_cantilever_beam.py
(lines 71-85, 115-131) on the closed catalogue."""
import numpy as np

from .forms import BeamResidual, LinearFunctional
from .function import Function, FunctionSpace


class EndpointMeasure:
    """ds_(100) of the example (:119-131): the end point x = L of the interval."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.dof = 2 * mesh.nel            # deflection DOF of the last node


def point_load(V: FunctionSpace, f: float, dss: EndpointMeasure) -> Function:
    """Nodal load vector of  dot(f, v) * dss."""
    F = Function(V)
    a = np.zeros(V.dim)
    a[dss.dof] = f
    F.vector[:] = a
    return F


def pdeRes(u, v, t, f, dss, E, width):
    """:77-79"""
    return BeamResidual(u, t, point_load(u.function_space, float(f), dss), E, width)


def compliance(u, f, dss):
    """:84-85  dot(f, u) * dss"""
    return LinearFunctional(point_load(u.function_space, float(f), dss), u)


def volume(t, width, L=None, others=()):
    """:81-82  t * width * L * dx  (the reference multiplies by L = 1 as well)"""
    mesh = t.function_space.mesh
    c = Function(t.function_space)
    c.vector[:] = width * (1.0 if L is None else L) * mesh.cell_lengths()
    return LinearFunctional(c, t, others)


def locate_dofs_at_point(V: FunctionSpace, x0: float):
    """The two Hermite DOFs (deflection, rotation) of the node at x0 (:157-160)."""
    node = int(np.argmin(np.abs(V.mesh.nodes - x0)))
    return [np.array([2 * node], dtype=np.int32), np.array([2 * node + 1], dtype=np.int32)]
