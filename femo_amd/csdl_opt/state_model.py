"""Mirror of femo/csdl_opt/state_model.py (218 lines) on the HIP engine.

Method names, parameter names, dict conventions (assign vs ``+=``) and the
state carried between calls (``self.dRdu, self.dRdf_dict, self.A, self.ksp,
self.dR, self.du``) follow the reference line by line; see the citations.
Values may be NumPy arrays (CSDL backend) or ``DeviceArray`` (stay in HBM).
"""
from femo_amd.fea.fea_hip import *                     # noqa: F401,F403  (state_model.py:1)
from femo_amd.fea.fea_hip import FEA
from femo_amd.fea.utils_hip import (DeviceArray, SparseMatrix, assembleMatrix, assembleSystem, assembleVector,
                                    computeMatVecProductBwd, computeMatVecProductFwd, computePartials,
                                    createFunction, getFuncArray, setUpKSP_MUMPS, update)
from femo_amd.csdl_opt._csdl_compat import Model, CustomImplicitOperation, custom
import numpy as np


def _on_device(d) -> bool:
    return any(isinstance(d[k], DeviceArray) for k in d)


class StateModel(Model):
    """state_model.py:7-38"""

    def initialize(self):
        self.parameters.declare('debug_mode', default=False)
        self.parameters.declare('fea', types=FEA)
        self.parameters.declare('state_name', types=str)
        self.parameters.declare('arg_name_list', types=list)

    def define(self):
        self.fea = self.parameters['fea']
        arg_name_list = self.parameters['arg_name_list']
        state_name = self.parameters['state_name']
        self.debug_mode = self.parameters['debug_mode']
        args_dict = dict()
        args_list = []
        for arg_name in arg_name_list:
            args_dict[arg_name] = self.fea.inputs_dict[arg_name]
            arg = self.declare_variable(arg_name,
                                        shape=(args_dict[arg_name]['shape'],),
                                        val=getFuncArray(args_dict[arg_name]['function']))
            args_list.append(arg)
            self.print_var(arg)

        e = StateOperation(fea=self.fea,
                           args_dict=args_dict,
                           state_name=state_name,
                           debug_mode=self.debug_mode)
        state = custom(*args_list, op=e)
        self.register_output(state_name, state)


class StateOperation(CustomImplicitOperation):
    """
    input: input variable
    output: state
    (state_model.py:41-218)
    """

    def initialize(self):
        self.parameters.declare('debug_mode')
        self.parameters.declare('fea')
        self.parameters.declare('args_dict')
        self.parameters.declare('state_name')

    def _banner(self, what):
        if self.debug_mode == True:
            print(str(self.state_name) + "=" * 40)
            print("CSDL: Running " + what + "...")
            print("=" * 40)

    def define(self):
        self.debug_mode = self.parameters['debug_mode']
        self.fea = self.parameters['fea']
        self.state_name = state_name = self.parameters['state_name']
        self.args_dict = args_dict = self.parameters['args_dict']
        self._banner("define()")

        for arg_name in args_dict:
            arg = args_dict[arg_name]
            self.add_input(arg_name,
                           shape=(arg['shape'],),)

        self.state = self.fea.states_dict[state_name]
        self.add_output(state_name,
                        shape=(self.state['shape'],),)
        self.declare_derivatives('*', '*')
        self.bcs = self.fea.bc
        self.linear = self.fea.linear_problem
        self.ksp = None

    def evaluate_residuals(self, inputs, outputs, residuals):
        """state_model.py:75-85: residual WITHOUT any BC treatment."""
        self._banner("evaluate_residuals()")

        for arg_name in inputs:
            arg = self.args_dict[arg_name]
            update(arg['function'], inputs[arg_name])
        update(self.state['function'], outputs[self.state_name])
        residuals[self.state_name] = assembleVector(self.state['residual_form'],
                                                    device=_on_device(inputs))

    def solve_residual_equations(self, inputs, outputs):
        """state_model.py:87-115"""
        self._banner("solve_residual_equations()")

        self.fea.opt_iter += 1
        for arg_name in inputs:
            arg = self.args_dict[arg_name]
            update(arg['function'], inputs[arg_name])
            if arg['record']:
                arg['recorder'].write_function(arg['function'],
                                               self.fea.opt_iter)

        update(self.state['function'], outputs[self.state_name])

        self.fea.solve(self.state['residual_form'],
                       self.state['function'],
                       self.bcs)

        outputs[self.state_name] = getFuncArray(self.state['function'], device=_on_device(inputs))
        if self.fea.record:
            self.state['recorder'].write_function(self.state['function'],
                                                  self.fea.opt_iter)

    def compute_derivatives(self, inputs, outputs, derivatives):
        """state_model.py:117-158: dRdu and dRdf WITHOUT BCs, A WITH BCs."""
        self._banner("compute_derivatives()")

        for arg_name in inputs:
            update(self.args_dict[arg_name]['function'], inputs[arg_name])
        update(self.state['function'], outputs[self.state_name])

        state = self.state
        args_dict = self.args_dict
        dR_du = state['dR_du']
        if dR_du == None:
            dR_du = computePartials(state['residual_form'], state['function'])
        # dRdu (no BCs, state_model.py:132) and A (BCs, state_model.py:149) come out of
        # ONE pass over the mesh below (assembleSystem(..., out_nobc=self.dRdu)).
        if getattr(self, 'dRdu', None) is None:
            self.dRdu = SparseMatrix(state['function'].function_space.mesh,
                                     symmetric=getattr(state['residual_form'], 'is_symmetric', False))
        dRdf_dict = dict()
        dR_df_list = state['dR_df_list']
        arg_list = state['arguments']
        old = getattr(self, 'dRdf_dict', {})
        for arg_ind in range(len(arg_list)):
            arg_name = arg_list[arg_ind]
            if dR_df_list == None:
                dRdf = assembleMatrix(computePartials(
                                        state['residual_form'],
                                        args_dict[arg_name]['function']),
                                      out=old.get(arg_name, {}).get('dRdf'))
            else:
                dRdf = dR_df_list[arg_ind]

            df = old[arg_name]['df'] if arg_name in old else createFunction(args_dict[arg_name]['function'])
            dRdf_dict[arg_name] = dict(dRdf=dRdf, df=df)

        self.dRdf_dict = dRdf_dict
        self.A, _ = assembleSystem(dR_du,
                                   state['residual_form'],
                                   bcs=self.bcs, rhs=False, out=getattr(self, 'A', None),
                                   out_nobc=self.dRdu)
        self.dR = self.state['d_residual']
        self.du = self.state['d_state']
        if self.linear is True:
            self.ksp = setUpKSP_MUMPS(self.A)

    def _bc_filter(self, values):
        """Only with ``fea.consistent_bc_partials``: zero the Dirichlet entries
        (not reference behaviour; used to verify against finite differences)."""
        if not self.fea.consistent_bc_partials or not self.bcs:
            return values
        dofs = np.unique(np.concatenate([bc.dofs for bc in self.bcs]))
        host = np.array(values, dtype=np.float64, copy=True)
        host[dofs] = 0.0
        if isinstance(values, DeviceArray):
            values.vec.set(host)
            return values
        return host

    def compute_jacvec_product(self, inputs, outputs,
                               d_inputs, d_outputs, d_residuals, mode):
        """state_model.py:161-200: accumulate (+=) into whatever keys are present."""
        self._banner("compute_jacvec_product()" + "mode " + str(mode))

        ######################
        # Might be redundant #
        for arg_name in inputs:
            update(self.args_dict[arg_name]['function'], inputs[arg_name])
        update(self.state['function'], outputs[self.state_name])
        ######################
        state_name = self.state_name
        dev = _on_device(inputs)
        if mode == 'fwd':
            if state_name in d_residuals:
                if state_name in d_outputs:
                    update(self.du, d_outputs[state_name])
                    d_residuals[state_name] += computeMatVecProductFwd(
                            self.dRdu, self.du, device=dev)
                for arg_name in self.dRdf_dict:
                    if arg_name in d_inputs:
                        update(self.dRdf_dict[arg_name]['df'],
                               d_inputs[arg_name])
                        dRdf = self.dRdf_dict[arg_name]['dRdf']
                        d_residuals[state_name] += computeMatVecProductFwd(
                                dRdf, self.dRdf_dict[arg_name]['df'], device=dev)

        if mode == 'rev':
            if state_name in d_residuals:
                update(self.dR, self._bc_filter(d_residuals[state_name]))
                if state_name in d_outputs:
                    d_outputs[state_name] += computeMatVecProductBwd(
                            self.dRdu, self.dR, device=dev)
                for arg_name in self.dRdf_dict:
                    if arg_name in d_inputs:
                        dRdf = self.dRdf_dict[arg_name]['dRdf']
                        d_inputs[arg_name] += computeMatVecProductBwd(
                                dRdf, self.dR, device=dev)

    def apply_inverse_jacobian(self, d_outputs, d_residuals, mode):
        """state_model.py:202-218: overwrite semantics."""
        self._banner("apply_inverse_jacobian()" + "mode " + str(mode))

        state_name = self.state_name
        if mode == 'fwd':
            d_outputs[state_name] = self.fea.solveLinearFwd(
                            self.du, self.A, self.dR,
                            d_residuals[state_name],
                            self.ksp, device=isinstance(d_residuals[state_name], DeviceArray))
        else:
            d_residuals[state_name] = self.fea.solveLinearBwd(
                            self.dR, self.A, self.du,
                            d_outputs[state_name],
                            self.ksp, device=isinstance(d_outputs[state_name], DeviceArray))
