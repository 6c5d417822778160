"""The explicit output operators of femo's CSDL layer on the HIP engine.

Public surface = the reference's femo/csdl_opt/output_model.py: ``OutputModel`` /
``OutputOperation`` for scalar functionals (value + partials w.r.t. every argument) and
``OutputFieldModel`` / ``OutputFieldOperation`` for L2-projected fields (value only; the reference
leaves their derivatives undeclared, output_model.py:147).  Model parameters ``fea, output_name,
arg_name_list``; operation parameters ``fea, args_dict, output_name``.
"""
import numpy as np

from femo_amd.csdl_opt._common import declare_all, gather_arguments, push_functions, stays_on_device
from femo_amd.csdl_opt._csdl_compat import CustomExplicitOperation, Model, custom
from femo_amd.engine import lazy_results
from femo_amd.fea.fea_hip import FEA
from femo_amd.fea.utils_hip import assemble, computePartials, getFuncArray


class _OutputOperationBase(CustomExplicitOperation):
    registry = 'outputs_dict'          # which FEA dict holds the output entry

    def initialize(self):
        declare_all(self.parameters, [('fea',), ('args_dict',), ('output_name',)])

    def define(self):
        P = self.parameters
        self.fea, self.output_name, self.args_dict = P['fea'], P['output_name'], P['args_dict']
        for name, entry in self.args_dict.items():
            self.add_input(name, shape=(entry['shape'],))
        self.output = getattr(self.fea, self.registry)[self.output_name]
        self.output_size = self.output['shape']
        self.output_dim = self._dimension()
        self.add_output(self.output_name, shape=(self.output_size,))
        self._declare()

    def _dimension(self) -> int:
        return 1

    def _declare(self) -> None:
        pass


class OutputOperation(_OutputOperationBase):
    """Scalar functional J(arguments) (output_model.py:40-87)."""

    def _dimension(self) -> int:
        return 0 if self.output_size == 1 else 1        # 0: scalar, 1: field-valued form

    def _declare(self) -> None:
        self.declare_derivatives('*', '*')

    def compute(self, inputs, outputs):
        """output_model.py:69-75"""
        push_functions(self.args_dict, inputs)
        outputs[self.output_name] = np.array(assemble(self.output['form'], dim=self.output_dim))

    def compute_derivatives(self, inputs, derivatives):
        """One assembled partial per argument, rank = output rank + 1 (output_model.py:77-87)."""
        push_functions(self.args_dict, inputs)
        dev = stays_on_device(inputs)
        form = self.output['form']
        # smallest first: with asynchronous results the copies leave in this order, and the adjoint seed (dJ/du)
        # is what the backend needs next
        with lazy_results(self.fea.async_results):
            for name, entry in sorted(self.args_dict.items(), key=lambda kv: kv[1]['shape']):
                derivatives[self.output_name, name] = assemble(computePartials(form, entry['function']),
                                                               dim=self.output_dim + 1, device=dev)


class OutputFieldOperation(_OutputOperationBase):
    """CG1 field obtained by L2 projection of the output's expression (output_model.py:122-159)."""
    registry = 'outputs_field_dict'

    def compute(self, inputs, outputs):
        push_functions(self.args_dict, inputs)
        out = self.output
        self.fea.projectFieldOutput(out['form'], out['func'])
        if out['record']:
            out['recorder'].write_function(out['func'], self.fea.opt_iter)
        with lazy_results(self.fea.async_results):
            outputs[self.output_name] = getFuncArray(out['func'], device=stays_on_device(inputs))


class _OutputModelBase(Model):
    operation = OutputOperation

    def initialize(self):
        declare_all(self.parameters, [('fea', dict(types=FEA)), ('output_name', dict(types=str)),
                                      ('arg_name_list', dict(types=list))])

    def define(self):
        P = self.parameters
        self.fea = P['fea']
        args_dict = gather_arguments(self.fea, P['arg_name_list'], allow_states=True)
        variables = [self.declare_variable(name, shape=(entry['shape'],), val=1.0)
                     for name, entry in args_dict.items()]
        op = self.operation(fea=self.fea, args_dict=args_dict, output_name=P['output_name'])
        self.print_var(self.register_output(P['output_name'], custom(*variables, op=op)))


class OutputModel(_OutputModelBase):
    """output_model.py:7-38"""
    operation = OutputOperation


class OutputFieldModel(_OutputModelBase):
    """output_model.py:90-120"""
    operation = OutputFieldOperation
