"""``FEA``: the per-PDE registry of femo on the HIP engine (reference: femo/fea/fea_dolfinx.py:70-234).

Same public attributes (``inputs_dict, states_dict, outputs_dict, outputs_field_dict, bc, PDE_SOLVER,
REPORT, custom_solve, opt_iter, initial_solve, initialize, record, recorder_path, linear_problem``), the
same registration methods and entry layouts, and the same solve entry points (``solve``,
``solveLinearFwd``, ``solveLinearBwd``, ``projectFieldOutput``).  Forms come from the closed catalogue
(``forms.py``, ``nonlinear_poisson.py``, ``beam.py``) instead of UFL; every dolfinx/PETSc call is the
``utils_hip`` function of the same name.  Like the reference module, this one re-exports the utility
layer so that run scripts can ``from ...fea_hip import *``.
"""
from __future__ import annotations

import os

import numpy as np

from .. import engine as E
from .utils_hip import *          # noqa: F401,F403  re-export (the reference does the same, fea_dolfinx.py:5)
from .utils_hip import (DeviceArray, DirichletBC, KSP, dirichletbc, getFuncArray, project, setFuncArray,
                        solveKSP_mumps, solveNonlinear, transpose)
from .forms import (ALPHA, BeamResidual, DerivativeForm, FieldExpression, Form, FunctionExpr, GradientMagnitude,
                    L2TrackingFunctional, LinearFunctional, NonlinearPoissonResidual, PoissonResidual, PowerExpr,
                    TestFunction, derivative, interiorResidual, outputForm, pdeRes)
from .function import Function, FunctionSpace
from .io import XDMFRecorder
from .mesh import (BeamMesh, Mesh, createIntervalMesh, createRectangleMesh, createUnitCubeMesh, createUnitSquareMesh,
                   findNodeIndices, locate_dofs_geometrical, meshSize)


class AbstractFEA(object):
    """The bare registry the reference keeps next to ``FEA`` (fea_dolfinx.py:20-67): named inputs, states
    with their residual forms and argument lists, outputs and a flat list of boundary conditions.  No
    shapes, recorders or solvers -- those live in ``FEA``."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.inputs_dict, self.states_dict, self.outputs_dict = {}, {}, {}
        self.bcs_list = []

    def add_strong_bc(self, bc):
        self.bcs_list.append(bc)

    def add_input(self, name, function):
        if name in self.inputs_dict:
            raise ValueError('name has already been used for an input')
        function.rename(name, name)
        self.inputs_dict[name] = {'function': function}

    def add_state(self, name, function, residual_form, *arguments):
        function.rename(name, name)
        self.states_dict[name] = {'function': function, 'residual_form': residual_form, 'arguments': arguments}

    def add_output(self, name, form, *arguments):
        self.outputs_dict[name] = {'form': form, 'arguments': arguments}


class FEA(object):
    """Registers inputs, states and outputs of one PDE problem and owns its nonlinear and linearised
    solves."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.inputs_dict, self.states_dict = {}, {}
        self.outputs_dict, self.outputs_field_dict = {}, {}
        self.bc = []
        # solver switches read by solve() / StateOperation.define (fea_dolfinx.py:87-98)
        self.PDE_SOLVER, self.REPORT = "Newton", True
        self.ubc = None
        self.custom_solve, self.initial_solve = None, True
        self.opt_iter = 0
        self.initialize = False
        self.record, self.recorder_path = False, "records"
        self.linear_problem = False
        # additions (not in the reference)
        self.consistent_bc_partials = False   # zero Dirichlet rows of dR/du, dR/df in the jac-vec products
        self.reference_fwd_bug = False        # reproduce solveLinearFwd's zeros (fea_dolfinx.py:192-206)
        self.reload_in_jacvec = False         # re-send inputs/state in compute_jacvec_product (state_model.py:168-173)
        # True: arrays the operator methods return may still be in flight (engine.lazy_results); the backend that
        # sets it calls engine.host_wait / host_sync before reading them with code of its own (csdl_opt/simulator.py)
        self.async_results = False

    # ------------------------------------------------------------------ registration ----
    def _recorded(self, name, record, **fields):
        entry = dict(fields)
        entry['recorder'] = self.createRecorder(name, record)
        entry['record'] = record
        return entry

    def add_input(self, name, function, init_val=1.0, record=False):
        """Every DOF of ``function`` is set to ``init_val``; duplicate names are an error
        (fea_dolfinx.py:100-110)."""
        if name in self.inputs_dict:
            raise ValueError('name has already been used for an input')
        function.x.array[:] = init_val
        self.inputs_dict[name] = self._recorded(name, record, function=function,
                                                function_space=function.function_space,
                                                shape=len(getFuncArray(function)))

    def add_state(self, name, function, residual_form, arguments, dR_du=None, dR_df_list=None, record=False):
        """Besides the state Function the entry carries two work Functions of the same space for the
        linearised solves (fea_dolfinx.py:112-127)."""
        V = function.function_space
        self.states_dict[name] = self._recorded(
            name, record, function=function, residual_form=residual_form, function_space=V,
            shape=len(getFuncArray(function)), d_residual=Function(V), d_state=Function(V),
            dR_du=dR_du, dR_df_list=dR_df_list, arguments=arguments)

    def add_output(self, name, type, form, arguments):
        """Scalar functional + its symbolic partials per argument (fea_dolfinx.py:129-146).  The
        reference's type='field' branch calls an undefined helper (:131); use add_field_output."""
        if type == 'field':
            raise NotImplementedError("add_output(type='field') is broken in the reference (undefined "
                                      "getFormArray); use add_field_output")
        if type != 'scalar':
            raise ValueError(f"unknown output type {type!r}")
        partials = []
        for arg in arguments:
            owner = self.inputs_dict if arg in self.inputs_dict else self.states_dict
            if arg not in owner:
                raise KeyError(f"output argument {arg!r} is neither an input nor a state")
            partials.append(derivative(form, owner[arg]['function']))
        self.outputs_dict[name] = dict(form=form, shape=1, arguments=arguments, partials=partials)

    def add_field_output(self, name, form, arguments, record=False):
        """CG1 field obtained by L2 projection of ``form`` (fea_dolfinx.py:148-161)."""
        space = self.mesh.field_space() if hasattr(self.mesh, "field_space") else FunctionSpace(self.mesh, ("CG", 1))
        func = Function(space)
        self.outputs_field_dict[name] = self._recorded(name, record, form=form, func=func,
                                                       shape=len(getFuncArray(func)), arguments=arguments,
                                                       partials=[])

    def add_exact_solution(self, Expression, function_space):
        """Interpolant of ``Expression().eval`` (fea_dolfinx.py:163-167)."""
        f_ex = Function(function_space)
        f_ex.interpolate(Expression().eval)
        return f_ex

    def add_strong_bc(self, ubc, locate_BC_list, function_space=None):
        """One Dirichlet condition per located DOF set (fea_dolfinx.py:169-176)."""
        self.bc.extend(dirichletbc(ubc, dofs, function_space) for dofs in locate_BC_list)

    # ------------------------------------------------------------------------ solves ----
    def solve(self, res, func, bc):
        """R(func) = 0 by the user hook or by solveNonlinear(PDE_SOLVER) (fea_dolfinx.py:178-189)."""
        if self.custom_solve is not None and self.initial_solve == True:
            self.custom_solve(res, func, bc, self.REPORT)
        else:
            solveNonlinear(res, func, bc, self.PDE_SOLVER, self.REPORT, self.initialize)

    def _linear_solve(self, operator, rhs_fn, rhs_values, sol_fn, ksp, device):
        setFuncArray(rhs_fn, rhs_values)
        sol_fn.vector.set(0.0)
        if ksp is None:
            solveKSP_mumps(operator, rhs_fn.vector, sol_fn.vector)
        else:
            ksp.solve(rhs_fn.vector, sol_fn.vector)
        with E.lazy_results(self.async_results):
            return getFuncArray(sol_fn, device=device)

    def solveLinearFwd(self, du, A, dR, dR_array, ksp=None, device=False):
        """du = A^-1 dR.  The reference swaps right-hand side and solution and therefore returns
        zeros (fea_dolfinx.py:192-206); the documented intent is
        implemented, ``reference_fwd_bug = True`` restores the zeros."""
        if self.reference_fwd_bug:
            setFuncArray(dR, dR_array)
            du.vector.set(0.0)
            return getFuncArray(du, device=device)
        return self._linear_solve(A, dR, dR_array, du, ksp, device)

    def solveLinearBwd(self, dR, A, du, du_array, ksp=None, device=False):
        """dR = A^-T du (fea_dolfinx.py:208-222).  With a cached ``ksp`` (linear_problem) the
        reference solves with A itself; here the cached object solves the same (symmetric) system."""
        return self._linear_solve(transpose(A), du, du_array, dR, ksp, device)

    def projectFieldOutput(self, form, func):
        """fea_dolfinx.py:224-225"""
        project(form, func, lump_mass=False)

    def createRecorder(self, name, record=False):
        """fea_dolfinx.py:228-234"""
        if not (record or self.record):
            return None
        # partitioned meshes: one file set per rank (every rank writes its own piece)
        local = getattr(self.mesh, "local", None)
        suffix = f"_rank{local.rank}" if local is not None and local.nranks > 1 else ""
        recorder = XDMFRecorder(os.path.join(self.recorder_path, "record_" + name + suffix + ".xdmf"))
        recorder.write_mesh(self.mesh)
        return recorder
