"""Mirror of femo/csdl_opt/fea_model.py:5-38: one StateModel per state and one
OutputModel per output of every FEA in the list; sub-model names
'{name}_state_model' / '{name}_output_model'."""
from femo_amd.csdl_opt._csdl_compat import Model
from femo_amd.csdl_opt.state_model import StateModel
from femo_amd.csdl_opt.output_model import OutputModel, OutputFieldModel


class FEAModel(Model):
    # The reference hard-codes debug_mode=True for every StateModel (fea_model.py:15),
    # which prints a banner per operator call; default to quiet, opt in per class/instance.
    debug_mode = False

    def initialize(self):
        self.parameters.declare('fea')

    def define(self):
        self.fea_list = fea_list = self.parameters['fea']
        if not isinstance(fea_list, (list, tuple)):
            raise TypeError("FEAModel(fea=[...]) takes a list of FEA objects (fea_model.py:10-11)")
        for fea in fea_list:
            for state_name in fea.states_dict:
                arg_name_list_state = fea.states_dict[state_name]['arguments']
                state_model = StateModel(fea=fea,
                                         debug_mode=self.debug_mode,
                                         state_name=state_name,
                                         arg_name_list=arg_name_list_state)

                self.add(state_model,
                         name='{}_state_model'.format(state_name))

            for output_name in fea.outputs_dict:
                arg_name_list_output = fea.outputs_dict[output_name]['arguments']
                output_model = OutputModel(fea=fea,
                                           output_name=output_name,
                                           arg_name_list=arg_name_list_output)

                self.add(output_model,
                         name='{}_output_model'.format(output_name))

            for output_name in fea.outputs_field_dict:
                arg_name_list_output = fea.outputs_field_dict[output_name]['arguments']
                output_model = OutputFieldModel(fea=fea,
                                                output_name=output_name,
                                                arg_name_list=arg_name_list_output)

                self.add(output_model,
                         name='{}_output_model'.format(output_name))
