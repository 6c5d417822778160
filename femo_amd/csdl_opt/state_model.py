"""The implicit PDE-state operator of femo's CSDL layer on the HIP engine.

Public surface = the reference's femo/csdl_opt/state_model.py: ``StateModel`` (parameters
``debug_mode, fea, state_name, arg_name_list``) and ``StateOperation`` (parameters ``debug_mode, fea,
args_dict, state_name``; methods ``define, evaluate_residuals, solve_residual_equations,
compute_derivatives, compute_jacvec_product, apply_inverse_jacobian``), with the same dict
conventions: assign for residuals / states / inverse-Jacobian results, ``+=`` into whatever keys are
present for Jacobian-vector products.  Between calls the operation keeps ``dRdu`` (no Dirichlet
rows eliminated), ``dRdf_dict``, ``A`` (eliminated), ``ksp``, ``dR``, ``du`` exactly like the
reference (state_model.py:117-158), so the backend's call order is unchanged.  Values may be NumPy
arrays (CSDL backend) or ``DeviceArray`` (stay in HBM).
"""
import numpy as np

from femo_amd.csdl_opt._common import (declare_all, gather_arguments, push_functions, stays_on_device,
                                       traced)
from femo_amd.csdl_opt._csdl_compat import CustomImplicitOperation, Model, custom
from femo_amd.engine import deferred_uploads, host_wait, lazy_results
from femo_amd.fea.fea_hip import FEA
from femo_amd.fea.forms import BackendForm
from femo_amd.fea.utils_hip import (DeviceArray, SparseMatrix, addMatVecProductBwd, addMatVecProductFwd,
                                    assembleMatrix, assembleSystem, assembleVector, computePartials,
                                    createFunction, getFuncArray, setUpKSP_MUMPS, update,
                                    KSP_OPTIONS)


import os as _os
_EARLY_DEFAULT = 'FEMO_NO_EARLY' not in _os.environ      # A/B switch for the early linearisation (read once)


class StateModel(Model):
    """Declares one variable per argument (initialised from the argument's Function) and registers
    the state as the output of a ``StateOperation`` (state_model.py:7-38)."""

    def initialize(self):
        declare_all(self.parameters, [('debug_mode', dict(default=False)), ('fea', dict(types=FEA)),
                                      ('state_name', dict(types=str)), ('arg_name_list', dict(types=list))])

    def define(self):
        P = self.parameters
        self.fea, self.debug_mode = P['fea'], P['debug_mode']
        args_dict = gather_arguments(self.fea, P['arg_name_list'], allow_states=False)
        variables = []
        for name, entry in args_dict.items():
            var = self.declare_variable(name, shape=(entry['shape'],), val=getFuncArray(entry['function']))
            self.print_var(var)
            variables.append(var)
        operation = StateOperation(fea=self.fea, args_dict=args_dict, state_name=P['state_name'],
                                   debug_mode=self.debug_mode)
        self.register_output(P['state_name'], custom(*variables, op=operation))


class StateOperation(CustomImplicitOperation):
    """R(arguments, state) = 0 with FE assembly and solves on the GPU (state_model.py:41-218)."""

    def initialize(self):
        declare_all(self.parameters, [('debug_mode',), ('fea',), ('args_dict',), ('state_name',)])

    @traced()
    def define(self):
        P = self.parameters
        self.debug_mode, self.fea = P['debug_mode'], P['fea']
        self.state_name, self.args_dict = P['state_name'], P['args_dict']
        self.state = self.fea.states_dict[self.state_name]
        for name, entry in self.args_dict.items():
            self.add_input(name, shape=(entry['shape'],))
        self.add_output(self.state_name, shape=(self.state['shape'],))
        self.declare_derivatives('*', '*')
        self.bcs, self.linear, self.ksp = self.fea.bc, self.fea.linear_problem, None

    # -- helpers -----------------------------------------------------------------------------
    def _load(self, inputs, outputs):
        """Arguments and current state -> Functions (state_model.py:81-84 and its repeats)."""
        push_functions(self.args_dict, inputs)
        update(self.state['function'], outputs[self.state_name])

    def _dirichlet_filtered(self, values):
        """Only with ``fea.consistent_bc_partials`` (not reference behaviour): zero the Dirichlet
        entries of the multiplier before the products, which turns the adjoint total into the exact
        reduced gradient (used to check against finite differences)."""
        if not (self.fea.consistent_bc_partials and self.bcs):
            return values
        dofs = np.unique(np.concatenate([bc.dofs for bc in self.bcs]))
        if isinstance(values, np.ndarray):
            host_wait(values)          # apply_inverse_jacobian may have returned it while its copy-out is in flight
        host = np.array(values, dtype=np.float64, copy=True)
        host[dofs] = 0.0
        if isinstance(values, DeviceArray):
            values.vec.set(host)
            return values
        return host

    # -- protocol ----------------------------------------------------------------------------
    @traced()
    def evaluate_residuals(self, inputs, outputs, residuals):
        """Residual vector without any Dirichlet treatment (state_model.py:75-85)."""
        self._load(inputs, outputs)
        with lazy_results(self.fea.async_results):
            residuals[self.state_name] = assembleVector(self.state['residual_form'], device=stays_on_device(inputs))

    @traced()
    def solve_residual_equations(self, inputs, outputs):
        """Nonlinear solve from the incoming state as initial guess (state_model.py:87-115)."""
        fea = self.fea
        fea.opt_iter += 1
        res = self.state['residual_form']
        # The state first, then the inputs as DEFERRED uploads (engine.deferred_uploads): for the catalogue forms the first
        # assembly pass orders its input-independent half -- matrix, K u', S A S -- in front of the wait for f, so that work
        # runs under the transfer.  Forms with a backend of their own (shell) and recorders read the inputs through other
        # entry points: they keep the synchronous upload.
        recording = any(self.args_dict[name]['record'] for name in inputs)
        defer = (fea.async_results and not isinstance(res, BackendForm) and not recording
                 and fea.custom_solve is None and getattr(fea, "deferred_uploads", True))
        update(self.state['function'], outputs[self.state_name])
        with deferred_uploads(defer):
            push_functions(self.args_dict, inputs)
            for name in inputs:
                entry = self.args_dict[name]
                if entry['record']:
                    entry['recorder'].write_function(entry['function'], fea.opt_iter)
            # Early linearisation (round 5).  For a form whose partials depend on the mesh and the Dirichlet set only
            # (`constant_partials`: linear Poisson) the assembly of dR/du, A, dR/df and S A S of the adjoint system -- what
            # compute_derivatives does after the solve (state_model.py:117-158) -- needs neither f nor u: it is issued HERE,
            # while f is still on its way to the device, and compute_derivatives of this cycle finds it done.  The same
            # kernels once per cycle, earlier: 2.1 ms of the 10 M-DOF cycle move under the upload.  Only with deferred uploads
            # (a backend that declared it honours asynchronous arrays) and never twice: a compute_derivatives without a
            # solve before it assembles as always.  Round 6: only once this operation HAS been asked for
            # its derivatives -- a forward-only evaluation (no compute_derivatives ever) neither assembles nor keeps them.
            self._early_done = False
            if defer and getattr(res, 'constant_partials', False) and getattr(fea, 'early_linearisation', _EARLY_DEFAULT) \
                    and getattr(self, '_derivatives_requested', False) \
                    and self.state['dR_du'] is None and self.state['dR_df_list'] is None:
                self._linearise()
                if hasattr(self.A, 'mat') and KSP_OPTIONS.get('pc') in ('bpx', 'jacobi'):
                    self.A.mat.prescale()
                self._early_done = True
            fea.solve(res, self.state['function'], self.bcs)
        with lazy_results(fea.async_results):
            outputs[self.state_name] = getFuncArray(self.state['function'], device=stays_on_device(inputs))
        if fea.record:
            self.state['recorder'].write_function(self.state['function'], fea.opt_iter)

    @traced()
    def compute_derivatives(self, inputs, outputs, derivatives):
        """Assembles and keeps: dRdu and dRdf[arg] with NO Dirichlet elimination, A with it
        (state_model.py:117-158).  dRdu and A come out of one pass over the mesh."""
        self._load(inputs, outputs)
        self._derivatives_requested = True                 # from the next solve on the constant partials are assembled early
        if getattr(self, '_early_done', False):            # assembled under the upload of this cycle's input (solve_residual_equations)
            # valid only because these partials depend on neither the inputs nor the state
            assert getattr(self.state['residual_form'], 'constant_partials', False), "early linearisation of a form whose partials are not constant"
            self._early_done = False
            return
        self._linearise()

    def _linearise(self):
        state, res = self.state, self.state['residual_form']
        dR_du = state['dR_du'] if state['dR_du'] is not None else computePartials(res, state['function'])
        if getattr(self, 'dRdu', None) is None:
            if hasattr(res, 'new_matrix'):                      # forms with a backend of their own (shell)
                self.dRdu = res.new_matrix()
            else:
                self.dRdu = SparseMatrix(state['function'].function_space.mesh,
                                         symmetric=getattr(res, 'is_symmetric', False))
        previous = getattr(self, 'dRdf_dict', {})
        self.dRdf_dict = {}
        for k, name in enumerate(state['arguments']):
            arg_fn = self.args_dict[name]['function']
            if state['dR_df_list'] is None:
                dRdf = assembleMatrix(computePartials(res, arg_fn), out=previous.get(name, {}).get('dRdf'))
            else:
                dRdf = state['dR_df_list'][k]                 # user-supplied matrices (fea_dolfinx.py:112-127)
            df = previous[name]['df'] if name in previous else createFunction(arg_fn)
            self.dRdf_dict[name] = dict(dRdf=dRdf, df=df)
        self.A, _ = assembleSystem(dR_du, res, bcs=self.bcs, rhs=False, out=getattr(self, 'A', None),
                                   out_nobc=self.dRdu)
        self.dR, self.du = state['d_residual'], state['d_state']
        if self.linear is True:
            self.ksp = setUpKSP_MUMPS(self.A)                  # solver object reused for every right-hand side

    @traced()
    def compute_jacvec_product(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        """fwd: d_residuals += dRdu du + sum dRdf df.   rev: d_outputs += dRdu^T dR,
        d_inputs[arg] += dRdf^T dR.  Keys that are absent are skipped (state_model.py:161-200)."""
        # The reference re-sends inputs and state here and calls it "might be redundant"
        # (state_model.py:168-173): the products only use the matrices compute_derivatives kept, so
        # the result does not depend on it.  Kept behind fea.reload_in_jacvec (default off): with
        # pageable arrays it would be 0.5 GB over PCIe per call on the 10 M-DOF cube.
        if getattr(self.fea, 'reload_in_jacvec', False):
            self._load(inputs, outputs)
        u = self.state_name
        if u not in d_residuals:
            return
        if mode == 'fwd':
            if u in d_outputs:
                update(self.du, d_outputs[u])
                d_residuals[u] = addMatVecProductFwd(d_residuals[u], self.dRdu, self.du)
            for name, pair in self.dRdf_dict.items():
                if name in d_inputs:
                    update(pair['df'], d_inputs[name])
                    d_residuals[u] = addMatVecProductFwd(d_residuals[u], pair['dRdf'], pair['df'])
        elif mode == 'rev':
            update(self.dR, self._dirichlet_filtered(d_residuals[u]))
            if u in d_outputs:
                d_outputs[u] = addMatVecProductBwd(d_outputs[u], self.dRdu, self.dR)
            for name, pair in self.dRdf_dict.items():
                if name in d_inputs:
                    d_inputs[name] = addMatVecProductBwd(d_inputs[name], pair['dRdf'], self.dR)

    @traced()
    def apply_inverse_jacobian(self, d_outputs, d_residuals, mode):
        """fwd: d_outputs = A^-1 d_residuals;  rev: d_residuals = A^-T d_outputs (state_model.py:202-218)."""
        u = self.state_name
        if mode == 'fwd':
            seed = d_residuals[u]
            d_outputs[u] = self.fea.solveLinearFwd(self.du, self.A, self.dR, seed, self.ksp,
                                                   device=isinstance(seed, DeviceArray))
        else:
            seed = d_outputs[u]
            d_residuals[u] = self.fea.solveLinearBwd(self.dR, self.A, self.du, seed, self.ksp,
                                                     device=isinstance(seed, DeviceArray))
