"""Mirror of femo/fea/fea_dolfinx.py: the ``FEA`` registry on the HIP engine.

Same attributes, method names, dict layouts and call order as the reference's
``FEA`` (fea_dolfinx.py:70-234); forms come from the closed catalogue in
``forms.py`` instead of UFL, and every dolfinx/PETSc call is replaced by the
``utils_hip`` function of the same name.
"""
from __future__ import annotations

import os

import numpy as np

from .utils_hip import *          # noqa: F401,F403  (the reference star-imports its utils too, fea_dolfinx.py:5)
from .utils_hip import (DeviceArray, DirichletBC, KSP, dirichletbc, getFuncArray, project, setFuncArray,
                        solveKSP_mumps, solveNonlinear, transpose)
from .forms import (ALPHA, DerivativeForm, FieldExpression, Form, FunctionExpr, GradientMagnitude,
                    L2TrackingFunctional, NonlinearPoissonResidual, PoissonResidual, PowerExpr, TestFunction,
                    derivative, interiorResidual, outputForm, pdeRes)
from .function import Function, FunctionSpace
from .mesh import (BeamMesh, Mesh, createIntervalMesh, createUnitCubeMesh, createUnitSquareMesh,
                   locate_dofs_geometrical)


class _NullRecorder:
    """XDMF recorders (fea_dolfinx.py:228-234) are out of scope:
    ``record=True`` writes raw ``.npy`` snapshots instead."""

    def __init__(self, path: str):
        self.path = path
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)

    def write_mesh(self, mesh) -> None:
        np.savez(self.path + "_mesh.npz", x=mesh.x, conn=mesh.conn)

    def write_function(self, function, t=0) -> None:
        np.save(f"{self.path}_{int(t):05d}.npy", function.vector.getArray())


class FEA(object):
    """
    The class of the FE wrapper: registers inputs / states / outputs of one PDE
    and provides the nonlinear and the linearised forward / transposed solves.
    (fea_dolfinx.py:70-234)
    """

    def __init__(self, mesh):
        self.mesh = mesh

        self.inputs_dict = dict()
        self.states_dict = dict()
        self.outputs_dict = dict()
        self.outputs_field_dict = dict()
        self.bc = []

        self.PDE_SOLVER = "Newton"
        self.REPORT = True

        self.ubc = None
        self.custom_solve = None

        self.opt_iter = 0
        self.initial_solve = True
        self.initialize = False
        self.record = False
        self.recorder_path = "records"
        self.linear_problem = False
        # not in the reference: zero the Dirichlet rows of dR/du and dR/df in the
        # jac-vec products (the reference keeps them, SURVEY.md section 0 finding 5)
        self.consistent_bc_partials = False

    def add_input(self, name, function, init_val=1.0, record=False):
        """fea_dolfinx.py:100-110"""
        if name in self.inputs_dict:
            raise ValueError('name has already been used for an input')
        function.x.array[:] = init_val
        self.inputs_dict[name] = dict(
            function=function,
            function_space=function.function_space,
            shape=len(getFuncArray(function)),
            recorder=self.createRecorder(name, record),
            record=record
        )

    def add_state(self, name, function, residual_form, arguments,
                  dR_du=None, dR_df_list=None, record=False):
        """fea_dolfinx.py:112-127"""
        self.states_dict[name] = dict(
            function=function,
            residual_form=residual_form,
            function_space=function.function_space,
            shape=len(getFuncArray(function)),
            d_residual=Function(function.function_space),
            d_state=Function(function.function_space),
            dR_du=dR_du,
            dR_df_list=dR_df_list,
            arguments=arguments,
            recorder=self.createRecorder(name, record),
            record=record
        )

    def add_output(self, name, type, form, arguments):
        """fea_dolfinx.py:129-146.  type='field' is broken in the reference
        (undefined getFormArray, :131); it raises here as well."""
        if type == 'field':
            raise NotImplementedError("add_output(type='field') calls an undefined helper in the reference; "
                                      "use add_field_output")
        elif type == 'scalar':
            shape = 1
        else:
            raise ValueError(f"unknown output type {type!r}")
        partials = []
        for argument in arguments:
            if argument in self.inputs_dict:
                partial = derivative(form, self.inputs_dict[argument]['function'])
            elif argument in self.states_dict:
                partial = derivative(form, self.states_dict[argument]['function'])
            else:
                raise KeyError(f"output argument {argument!r} is neither an input nor a state")
            partials.append(partial)
        self.outputs_dict[name] = dict(
            form=form,
            shape=shape,
            arguments=arguments,
            partials=partials,
        )

    def add_field_output(self, name, form, arguments, record=False):
        """fea_dolfinx.py:148-161: a CG1 field obtained by L2 projection of ``form``
        (a catalogue FieldExpression instead of a UFL expression)."""
        V = FunctionSpace(self.mesh, ("CG", 1))
        output_func = Function(V)
        partials = []
        self.outputs_field_dict[name] = dict(
            form=form,
            func=output_func,
            shape=len(getFuncArray(output_func)),
            arguments=arguments,
            partials=partials,
            recorder=self.createRecorder(name, record),
            record=record
        )

    def add_exact_solution(self, Expression, function_space):
        """fea_dolfinx.py:163-167"""
        f_analytic = Expression()
        f_ex = Function(function_space)
        f_ex.interpolate(f_analytic.eval)
        return f_ex

    def add_strong_bc(self, ubc, locate_BC_list, function_space=None):
        """fea_dolfinx.py:169-176"""
        if function_space == None:
            for locate_BC in locate_BC_list:
                self.bc.append(dirichletbc(ubc, locate_BC))
        else:
            for locate_BC in locate_BC_list:
                self.bc.append(dirichletbc(ubc, locate_BC, function_space))

    def solve(self, res, func, bc):
        """
        Solve the PDE problem (fea_dolfinx.py:178-189)
        """
        solver_type = self.PDE_SOLVER
        report = self.REPORT
        initialize = self.initialize
        if self.custom_solve is not None and self.initial_solve == True:
            self.custom_solve(res, func, bc, report)
        else:
            solveNonlinear(res, func, bc, solver_type, report, initialize)

    def solveLinearFwd(self, du, A, dR, dR_array, ksp=None, device=False):
        """
        solve linear system dR = dR_du (A) * du  (fea_dolfinx.py:192-206)

        The reference passes (b=du, x=dR) to the solver and returns ``du``, i.e. zeros
        (SURVEY.md section 8 row a13).  This implements the documented intent
        du = A^{-1} dR; set ``FEA.reference_fwd_bug = True`` to get the zeros.
        """
        setFuncArray(dR, dR_array)
        du.vector.set(0.0)
        if not getattr(self, "reference_fwd_bug", False):
            if ksp is None:
                solveKSP_mumps(A, dR.vector, du.vector)
            else:
                ksp.solve(dR.vector, du.vector)
        du.vector.assemble()
        du.vector.ghostUpdate()
        return getFuncArray(du, device=device)

    def solveLinearBwd(self, dR, A, du, du_array, ksp=None, device=False):
        """
        solve linear system du = dR_du.T (A_T) * dR  (fea_dolfinx.py:208-222)
        """
        setFuncArray(du, du_array)

        dR.vector.set(0.0)
        if ksp is None:
            solveKSP_mumps(transpose(A), du.vector, dR.vector)
        else:
            ksp.solve(du.vector, dR.vector)
        dR.vector.assemble()
        dR.vector.ghostUpdate()
        return getFuncArray(dR, device=device)

    def projectFieldOutput(self, form, func):
        """fea_dolfinx.py:224-225"""
        project(form, func, lump_mass=False)

    def createRecorder(self, name, record=False):
        """fea_dolfinx.py:228-234"""
        recorder = None
        if record or self.record:
            recorder = _NullRecorder(self.recorder_path + "/record_" + name)
            recorder.write_mesh(self.mesh)
        return recorder
