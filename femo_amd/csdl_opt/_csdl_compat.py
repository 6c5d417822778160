"""Stand-ins for the parts of ``csdl`` the operators touch.

csdl / python_csdl_backend are unpinned git clones in the reference
(README.md:21-24) and are not installed here.  If ``import csdl`` works, the real
classes are re-exported; otherwise these stubs implement exactly the protocol
used at state_model.py:7-73, output_model.py:7-67 and fea_model.py:5-38:
``parameters.declare``, ``declare_variable``, ``create_input``,
``register_output``, ``add``, ``csdl.custom``, ``add_input``/``add_output``/
``declare_derivatives``.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np

try:  # pragma: no cover - not available in the build container
    import csdl as _csdl
    from csdl import CustomExplicitOperation, CustomImplicitOperation, Model  # noqa: F401
    custom = _csdl.custom
    HAVE_CSDL = True
except Exception:  # ModuleNotFoundError in this image
    HAVE_CSDL = False

    class _Parameters:
        def __init__(self):
            self._decl: Dict[str, dict] = {}
            self._val: Dict[str, Any] = {}

        def declare(self, name, default=None, types=None, **kw):
            self._decl[name] = dict(default=default, types=types)
            if name not in self._val:
                self._val[name] = default

        def update(self, kwargs):
            for k, v in kwargs.items():
                if k not in self._decl:
                    raise KeyError(f"parameter {k!r} was not declared")
                t = self._decl[k]["types"]
                if t is not None and v is not None and not isinstance(v, t):
                    raise TypeError(f"parameter {k!r} must be of type {t}, got {type(v)}")
                self._val[k] = v

        def __getitem__(self, k):
            return self._val[k]

    class Variable:
        def __init__(self, name, shape, val=None, kind="declared"):
            self.name, self.shape, self.kind = name, tuple(np.atleast_1d(shape).tolist()), kind
            self.val = val
            self.op = None

    class _Parametrized:
        def __init__(self, **kwargs):
            self.parameters = _Parameters()
            self.initialize()
            self.parameters.update(kwargs)

        def initialize(self):
            pass

    class Model(_Parametrized):
        def __init__(self, **kwargs):
            super().__init__(**kwargs)
            self.variables: Dict[str, Variable] = {}
            self.submodels: List[Tuple[str, "Model"]] = []
            self.operations: List[Any] = []
            self.design_variables: Dict[str, dict] = {}
            self.objective: Optional[dict] = None
            self._defined = False

        def define(self):
            pass

        def declare_variable(self, name, shape=(1,), val=1.0):
            v = Variable(name, shape, val, "declared")
            self.variables[name] = v
            return v

        def create_input(self, name, shape=(1,), val=1.0):
            v = Variable(name, shape, val, "input")
            self.variables[name] = v
            return v

        def register_output(self, name, var):
            if isinstance(var, Variable):
                var.name = name
                var.kind = "output"
                self.variables[name] = var
            return var

        def add(self, submodel, name=None, promotes=None):
            self.submodels.append((name or type(submodel).__name__, submodel))
            return submodel

        def add_design_variable(self, name, lower=None, upper=None, scaler=None):
            self.design_variables[name] = dict(lower=lower, upper=upper, scaler=scaler)

        def add_objective(self, name, scaler=1.0):
            self.objective = dict(name=name, scaler=scaler)

        def print_var(self, var):
            pass

        def connect(self, a, b):
            pass

    class _CustomOperation(_Parametrized):
        def __init__(self, **kwargs):
            super().__init__(**kwargs)
            self.input_meta: Dict[str, dict] = {}
            self.output_meta: Dict[str, dict] = {}
            self.derivatives_meta: List[Tuple[str, str]] = []

        def define(self):
            pass

        def add_input(self, name, shape=(1,), val=1.0):
            self.input_meta[name] = dict(shape=tuple(np.atleast_1d(shape).tolist()), val=val)

        def add_output(self, name, shape=(1,), val=1.0):
            self.output_meta[name] = dict(shape=tuple(np.atleast_1d(shape).tolist()), val=val)

        def declare_derivatives(self, of, wrt, **kw):
            self.derivatives_meta.append((of, wrt))

    class CustomExplicitOperation(_CustomOperation):
        pass

    class CustomImplicitOperation(_CustomOperation):
        pass

    _CURRENT_MODEL: List[Model] = []

    def custom(*args, op=None):
        """csdl.custom(*inputs, op=operation): runs op.define() and returns its output variable(s)."""
        op.define()
        op.arg_variables = list(args)
        outs = []
        for name, meta in op.output_meta.items():
            v = Variable(name, meta["shape"], meta["val"], "output")
            v.op = op
            outs.append(v)
        return outs[0] if len(outs) == 1 else tuple(outs)
