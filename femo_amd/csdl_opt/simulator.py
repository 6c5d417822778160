"""A small driver that plays python_csdl_backend.Simulator [ext] for FEAModel graphs.

It exists so the operators can be exercised (tests, smoke, bench) without CSDL:
``run()`` walks the operations in the order FEAModel added them and calls
``solve_residual_equations`` / ``compute``;
``compute_totals`` does the reverse sweep of section 3.3:

    explicit op :  adj[arg] += seed * dJ/darg           (compute_derivatives)
    implicit op :  lam = (dR/du)^-T adj[u]               (apply_inverse_jacobian 'rev')
                   adj[f] -= (dR/df)^T lam               (compute_jacvec_product 'rev')

Variables are promoted by bare name, as the reference's run scripts rely on
(run_poisson_opt.py:168-176); ``sim['l2_functional_output_model.l2_functional']``
style addresses resolve to the same store.  ``device=True`` keeps every value
in HBM (``DeviceArray``) instead of NumPy arrays.

Host mode (``device=False``): like a CSDL backend the driver owns the storage of its variables, and
it keeps them in pinned blocks of the engine (``engine.pinned_empty``): what the operators receive are
read-only NumPy views of those blocks, what they return (read-only pinned arrays as well) is adopted by
reference.  The engine can therefore tell that an array it is handed again is still the exact copy of a
device vector and moves it over PCIe once (include/femo_hip.h, "host memory").  The reverse sweep
accumulates in place: the adjoint right-hand side is negated (one pass over a state-sized array), so that
``d_inputs[arg] += dRdf^T psi`` lands directly in the array that already holds the explicit partial.

Asynchronous results (``async_results=True``, host mode with pinned storage): the driver tells the FEA objects that
it honours the contract of ``engine.lazy_results`` -- arrays returned by operator methods may still be on their
way down; it never reads them with NumPy before ``engine.host_wait`` (``sim[name]``) or ``engine.host_sync`` (end of
``compute_totals``), and everything it does with them in between goes through library calls, which wait by
themselves.  The 477 MB of dJ/df then travel while the adjoint system is assembled and solved.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import sys

import numpy as np

from femo_amd import engine as E
from femo_amd.csdl_opt._csdl_compat import CustomExplicitOperation, CustomImplicitOperation
from femo_amd.engine import DeviceArray, Vec
from femo_amd.fea.utils_hip import get_context


def _readonly(a: np.ndarray) -> np.ndarray:
    v = a.view()
    v.flags.writeable = False
    return v


class Simulator:
    def __init__(self, model, device: bool = False, pinned: bool = True, async_results: bool = True):
        self.model = model
        self.device = device
        self.pinned = pinned          # host mode: variables in pinned engine blocks (False: plain pageable NumPy arrays)
        self.async_results = bool(async_results) and pinned and not device
        self.values: Dict[str, object] = {}
        self.ops: List[Tuple[str, object]] = []          # (submodel name, operation)
        self._build(model)

    # ------------------------------------------------------------- graph ----
    def _build(self, model) -> None:
        model.define()
        for name, var in model.variables.items():
            if var.kind == "input":
                self._store(name, self._initial(var))
        for sub_name, sub in model.submodels:
            sub.define()
            for name, var in sub.variables.items():
                if var.kind == "declared" and name not in self.values:
                    self._store(name, self._initial(var))
                elif var.kind == "output" and var.op is not None:
                    self.ops.append((sub_name, var.op))
                    if getattr(var.op, "fea", None) is not None:
                        var.op.fea.async_results = self.async_results
                    for oname, meta in var.op.output_meta.items():
                        if oname not in self.values:
                            self._store(oname, np.full(meta["shape"], float(np.asarray(meta["val"]).ravel()[0])))

    @staticmethod
    def _initial(var) -> np.ndarray:
        val = np.asarray(var.val, dtype=np.float64)
        return np.broadcast_to(val, var.shape).astype(np.float64).copy() if val.shape != var.shape else val.copy()

    def _store(self, name: str, value) -> None:
        if self.device and not isinstance(value, DeviceArray):
            a = np.asarray(value, dtype=np.float64).ravel()
            cur = self.values.get(name)
            if isinstance(cur, DeviceArray) and cur.n == a.size:
                cur.vec.set(a)
                return
            value = DeviceArray(Vec(get_context(), a.size).set(a))
        elif not self.device:
            if isinstance(value, DeviceArray):
                value = value.numpy()
            a = np.ascontiguousarray(value, dtype=np.float64).ravel()
            if not self.pinned:
                # a driver with pageable storage of its own: one persistent array per variable, written in place (what a
                # backend that preallocates its variable vectors does); the engine pins such an array in place the
                # second time it is handed over (engine._note_caller_array)
                store = self.__dict__.setdefault("_storage", {})
                w = store.get(name)
                if w is None or w.size != a.size:
                    w = store[name] = np.empty(a.size)
                if w is not a and not np.shares_memory(w, a):
                    E.host_copy(w, np.ascontiguousarray(E.host_wait(a)))
                value = w
            elif not a.flags.writeable and E.is_pinned(a):
                value = a                                   # a result of the engine (or engine.pinned_array): adopt
            else:
                store = self.__dict__.setdefault("_storage", {})
                w = store.get(name)
                if w is None or w.size != a.size:
                    w = store[name] = E.pinned_empty(a.size)
                E.host_copy(w, a)                           # the driver's own storage; announces the write
                value = _readonly(w)
        self.values[name] = value

    @staticmethod
    def _key(name: str) -> str:
        return name.split(".")[-1]

    def __getitem__(self, name: str):
        """NumPy value of a variable.  Host mode returns the stored (read-only) array itself."""
        v = self.values[self._key(name)]
        return v.numpy() if isinstance(v, DeviceArray) else E.host_wait(v)

    def __setitem__(self, name: str, value) -> None:
        self._store(self._key(name), value)

    def device_value(self, name: str):
        return self.values[self._key(name)]

    # --------------------------------------------------------------- run ----
    def run(self) -> None:
        for _, op in self.ops:
            inputs = {k: self.values[k] for k in op.input_meta}
            if isinstance(op, CustomImplicitOperation):
                outputs = {k: self.values[k] for k in op.output_meta}
                op.solve_residual_equations(inputs, outputs)
            else:
                outputs = {}
                op.compute(inputs, outputs)
            for k, v in outputs.items():
                if self.device and isinstance(v, DeviceArray):
                    self.values[k] = v
                else:
                    self._store(k, np.atleast_1d(np.asarray(v, dtype=np.float64)))

    # ------------------------------------------------------------ totals ----
    def _zeros_like(self, name: str, role: str = ""):
        """Zero array shaped like variable ``name``.  Device arrays are pooled per
        (name, role) and re-zeroed, so a sweep does not allocate."""
        v = self.values[name]
        if isinstance(v, DeviceArray):
            pool = self.__dict__.setdefault("_pool", {})
            a = pool.get((name, role))
            if a is None or a.n != v.n:
                a = DeviceArray.zeros(get_context(), v.n)
                pool[(name, role)] = a
            else:
                a.vec.fill(0.0)
            return a
        return np.zeros_like(np.asarray(v, dtype=np.float64))

    def _result(self, key, val):
        """What compute_totals hands out.  NumPy values are the caller's.  Device values are copied
        out of the sweep's pooled work arrays into a per-(of, wrt) result buffer that is reused only
        when the caller no longer holds the previous result."""
        if not isinstance(val, DeviceArray):
            return val
        pool = self.__dict__.setdefault("_results", {})
        out = pool.get(key)
        if out is None or out.n != val.n or sys.getrefcount(out) > 3:      # pool + local + getrefcount's argument
            out = pool[key] = DeviceArray.zeros(get_context(), val.n)
        out.vec.copy_from(val.vec)
        return out

    @staticmethod
    def _own(a: np.ndarray, announce: bool = True) -> np.ndarray:
        """A writable array holding ``a``: the array itself, a writable alias of a result the engine
        returned read-only (the sweep received it and is its only holder), or a copy.  ``announce=False``:
        only library calls will write it until the sweep touches it (``engine.writable``)."""
        if a.flags.writeable:
            return a
        try:
            return E.writable(a, announce)
        except ValueError:
            w = E.pinned_empty(a.size)
            E.host_copy(w, np.ascontiguousarray(a))
            return w

    def compute_totals(self, of: Union[str, Sequence[str]], wrt: Union[str, Sequence[str]]):
        """Reverse-mode total derivatives of scalar outputs ``of`` w.r.t. ``wrt``."""
        single = isinstance(of, str) and isinstance(wrt, str)
        ofs = [of] if isinstance(of, str) else list(of)
        wrts = [wrt] if isinstance(wrt, str) else list(wrt)
        result = {}
        for o in ofs:
            o = self._key(o)
            if np.size(self[o]) != 1:
                raise NotImplementedError("compute_totals handles scalar outputs")
            adj: Dict[str, object] = {}
            seed_done = False
            for _, op in reversed(self.ops):
                inputs = {k: self.values[k] for k in op.input_meta}
                if isinstance(op, CustomExplicitOperation):
                    if o not in op.output_meta:
                        continue
                    derivatives = {}
                    op.compute_derivatives(inputs, derivatives)
                    for (oo, arg), val in derivatives.items():
                        if isinstance(val, DeviceArray):
                            if arg not in adj:
                                adj[arg] = self._zeros_like(arg, 'adj')
                            adj[arg] += val
                        else:
                            val = np.ascontiguousarray(val, dtype=np.float64).ravel()
                            if not self.pinned:
                                # the driver's own (persistent, pageable) adjoint storage for this argument
                                # (a small ring per slot: an array handed out by an earlier compute_totals that its
                                # caller still holds is never written again -- ADVICE round 3; arrays that come back
                                # into rotation keep their in-place pinning, engine._note_caller_array)
                                ring = self.__dict__.setdefault("_adj_storage", {}).setdefault((o, arg, len(adj)), [])
                                ring[:] = [a for a in ring if a.size == val.size]
                                w = next((a for a in ring if sys.getrefcount(a) <= 3), None)   # ring + generator variable + argument
                                if w is None:
                                    w = np.empty(val.size)
                                    if len(ring) < 4:
                                        ring.append(w)
                                E.host_copy(w, E.host_wait(val))
                                val = w
                            if arg not in adj:
                                adj[arg] = val                      # by reference: no pass over the array
                            else:
                                adj[arg] = self._own(adj[arg])
                                E.host_axpby(1.0, val, 1.0, adj[arg])
                    seed_done = True
                else:
                    (state,) = tuple(op.output_meta)
                    if state not in adj:
                        continue
                    outputs = {state: self.values[state]}
                    op.compute_derivatives(inputs, outputs, {})
                    if isinstance(adj[state], DeviceArray):
                        d_outputs = {state: adj.pop(state)}
                        d_residuals = {state: self._zeros_like(state, 'd_res')}
                        op.apply_inverse_jacobian(d_outputs, d_residuals, 'rev')
                        d_inputs = {k: self._zeros_like(k, 'd_in') for k in op.input_meta}
                        op.compute_jacvec_product(inputs, outputs, d_inputs, {}, d_residuals, 'rev')
                        for k, v in d_inputs.items():
                            if k not in adj:
                                adj[k] = self._zeros_like(k, 'adj')
                            adj[k].vec.axpy(-1.0, v.vec)
                    else:
                        # psi = -(dR/du)^-T adj[u]; then adj[arg] += (dR/darg)^T psi in place
                        rhs = adj.pop(state)
                        seed = self.__dict__.setdefault("_seed", {}).get(state)
                        if seed is None or seed.size != rhs.size:
                            seed = self._seed[state] = E.pinned_empty(rhs.size) if self.pinned else np.empty(rhs.size)
                        E.host_axpby(-1.0, np.ascontiguousarray(rhs), 0.0, seed)
                        d_outputs = {state: _readonly(seed)}
                        d_residuals = {state: np.zeros(0)}          # assigned by the operation
                        op.apply_inverse_jacobian(d_outputs, d_residuals, 'rev')
                        d_inputs = {}
                        for k in op.input_meta:
                            if k in adj:
                                d_inputs[k] = self._own(adj[k], announce=False)
                            elif self.pinned:
                                d_inputs[k] = E.pinned_empty(np.size(self.values[k]))
                                E.host_axpby(0.0, d_inputs[k], 0.0, d_inputs[k])
                            else:
                                d_inputs[k] = np.zeros(np.size(self.values[k]))
                        op.compute_jacvec_product(inputs, outputs, d_inputs, {}, d_residuals, 'rev')
                        for v in d_inputs.values():
                            if self.pinned:
                                E.host_touch(v)           # whatever wrote it: it mirrors no device vector now
                        adj.update(d_inputs)
            if not seed_done:
                raise KeyError(f"no operation produces {o!r}")
            for w in wrts:
                w = self._key(w)
                val = adj[w] if w in adj else self._zeros_like(w, 'adj')
                result[(o, w)] = self._result((o, w), val)
        if self.async_results:
            E.host_sync()
        if single:
            return result[(self._key(of), self._key(wrt))]
        return result

    def check_totals(self, of: str, wrt: str, step: float = 1e-6, n_dir: int = 3, seed: int = 0) -> dict:
        """Directional central finite differences vs the adjoint total (the idiom of
        run_aeroelasticity_static_wo_feedback.py:389-394, on random directions)."""
        g = np.asarray(self.compute_totals(of, wrt), dtype=np.float64).ravel()
        x0 = np.array(self[wrt], dtype=np.float64, copy=True)
        rng = np.random.default_rng(seed)
        out = dict(analytical=[], fd=[], rel_error=[])
        for _ in range(n_dir):
            d = rng.standard_normal(x0.shape)
            vals = []
            for s in (+1.0, -1.0):
                self[wrt] = x0 + s * step * d
                self.run()
                vals.append(float(np.asarray(self[of]).ravel()[0]))
            fd = (vals[0] - vals[1]) / (2 * step)
            an = float(g @ d)
            out["analytical"].append(an)
            out["fd"].append(fd)
            out["rel_error"].append(abs(an - fd) / max(abs(fd), 1e-300))
        self[wrt] = x0
        self.run()
        return out
