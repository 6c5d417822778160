"""``FEAModel(fea=[...])``: the composite CSDL model of femo (reference: femo/csdl_opt/fea_model.py).

For every FEA in the list it adds one sub-model per registered state, scalar output and field output,
named ``'<name>_state_model'`` / ``'<name>_output_model'`` -- the names the reference's run scripts
address (``sim['l2_functional_output_model.l2_functional']``, run_poisson_opt.py:239).
"""
from femo_amd.csdl_opt._csdl_compat import Model
from femo_amd.csdl_opt.output_model import OutputFieldModel, OutputModel
from femo_amd.csdl_opt.state_model import StateModel

# (FEA registry, sub-model class, name suffix, keyword of the registered name)
_WIRING = (
    ('states_dict', StateModel, 'state_model', 'state_name'),
    ('outputs_dict', OutputModel, 'output_model', 'output_name'),
    ('outputs_field_dict', OutputFieldModel, 'output_model', 'output_name'),
)


class FEAModel(Model):
    # The reference switches the per-call banners of every StateModel on (fea_model.py:15);
    # here they are opt-in: set ``FEAModel.debug_mode = True`` (class or instance).
    debug_mode = False

    def initialize(self):
        self.parameters.declare('fea')

    def define(self):
        self.fea_list = self.parameters['fea']
        if not isinstance(self.fea_list, (list, tuple)):
            raise TypeError("FEAModel(fea=[...]) takes a list of FEA objects")
        for fea in self.fea_list:
            for registry, model_cls, suffix, key in _WIRING:
                for name, entry in getattr(fea, registry).items():
                    kwargs = {'fea': fea, key: name, 'arg_name_list': list(entry['arguments'])}
                    if model_cls is StateModel:
                        kwargs['debug_mode'] = self.debug_mode
                    self.add(model_cls(**kwargs), name=f'{name}_{suffix}')
