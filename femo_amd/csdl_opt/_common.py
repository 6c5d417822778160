"""Shared plumbing of the CSDL operator mirrors (no counterpart file in the reference, which
repeats these idioms inline in state_model.py / output_model.py)."""
from __future__ import annotations

import functools
from typing import Dict, Iterable, Mapping, Tuple

from femo_amd.fea.utils_hip import DeviceArray, update


def declare_all(parameters, spec: Iterable[Tuple]) -> None:
    """spec rows: (name,) | (name, {declare kwargs})."""
    for row in spec:
        name, kw = row[0], (row[1] if len(row) > 1 else {})
        parameters.declare(name, **kw)


def push_functions(args_dict: Mapping[str, dict], values: Mapping[str, object]) -> None:
    """Write every value that has an entry in ``args_dict`` into its Function (host->device copy
    for NumPy values, device copy / no-op for DeviceArray): the reference's per-method
    ``for arg_name in inputs: update(...)`` loops (state_model.py:81-84, 94-96, 123-124, 171-172;
    output_model.py:70-72, 78-80, 150-152)."""
    for name in values:
        update(args_dict[name]['function'], values[name])


def stays_on_device(values: Mapping[str, object]) -> bool:
    return any(isinstance(v, DeviceArray) for v in values.values())


def gather_arguments(fea, names: Iterable[str], allow_states: bool) -> Dict[str, dict]:
    """Registry entries of the named arguments: inputs, and (for outputs) states as well."""
    found: Dict[str, dict] = {}
    for name in names:
        if name in fea.inputs_dict:
            found[name] = fea.inputs_dict[name]
        elif allow_states and name in fea.states_dict:
            found[name] = fea.states_dict[name]
        elif not allow_states:
            found[name] = fea.inputs_dict[name]          # KeyError for unknown names, as in the reference
    return found


def traced(label: str = None):
    """Optional banner per operator call (the reference prints one when debug_mode is on)."""
    def deco(method):
        @functools.wraps(method)
        def wrapper(self, *a, **k):
            if getattr(self, 'debug_mode', False) == True:
                extra = ""
                if 'mode' in k or (a and isinstance(a[-1], str)):
                    extra = "mode " + str(k.get('mode', a[-1] if a else ''))
                bar = "=" * 40
                print(f"{self.state_name}{bar}\nCSDL: Running {label or method.__name__}()...{extra}\n{bar}")
            return method(self, *a, **k)
        return wrapper
    return deco
