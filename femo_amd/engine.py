"""Thin object layer over the C-ABI handles (no numerics here).

Each class owns one opaque handle of ``libfemo_hip.so``; NumPy arrays cross the
boundary as plain pointers + sizes.  The dolfinx/PETSc objects these stand in
for are named per class.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import FemoError, H, SolveInfo, SolverOpts, check


PC_KINDS = {"jacobi": 0, "bpx": 1}     # include/femo_hip.h FEMO_PC_*


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _h(obj) -> Optional[H]:
    return None if obj is None else obj.handle


# ------------------------------------------------------------------ host memory ----
class _PinnedBlock:
    """One block of ``femo_host_alloc``; NumPy arrays built on it keep it alive (``ndarray.base``),
    the last reference returns it to the library's recycling list."""

    __slots__ = ("ptr", "n", "__array_interface__", "__weakref__")

    def __init__(self, n: int):
        lib = _lib.load()
        p = C.c_void_p()
        check(lib.femo_host_alloc(max(int(n), 1) * 8, C.byref(p)))
        self.ptr, self.n = p.value, int(n)
        self.__array_interface__ = dict(shape=(self.n,), typestr="<f8", data=(self.ptr, False), version=3)

    def __del__(self):
        p, self.ptr = self.ptr, None
        if p:
            try:
                _lib.load().femo_host_free(C.c_void_p(p))
            except Exception:
                pass


def pinned_empty(n: int) -> np.ndarray:
    """Writable fp64 array in pinned host memory (one DMA to / from the device).  The owner of such
    an array calls ``host_touch`` after writing to it; arrays the library *returns* are read-only
    views instead, so that it can trust them as exact copies of the device vector they came from."""
    return np.asarray(_PinnedBlock(n))


def pinned_array(values) -> np.ndarray:
    """Read-only pinned copy of ``values`` (what bench.py / a driver hands to the operators)."""
    a = _f64(values).ravel()
    out = pinned_empty(a.size)
    host_copy(out, a)
    out.flags.writeable = False
    return out


def is_pinned(a: np.ndarray) -> bool:
    return bool(_lib.load().femo_host_is_pinned(C.c_void_p(a.ctypes.data), a.nbytes))


def host_touch(a: np.ndarray) -> None:
    """Tell the library that ``a`` (pinned) was written on the host: it no longer mirrors a device vector."""
    check(_lib.load().femo_host_touch(C.c_void_p(a.ctypes.data)))


def pinned_full(n: int, value: float) -> np.ndarray:
    """READ-ONLY pinned array of n copies of ``value`` that the library remembers as that constant: sending it to a
    device vector is a fill kernel, not a transfer (initial guesses and other constant fields of a driver)."""
    a = pinned_empty(n)
    check(_lib.load().femo_host_fill(C.c_void_p(a.ctypes.data), int(n), float(value)))
    a.flags.writeable = False
    return a


def host_wait(a: np.ndarray) -> np.ndarray:
    """Block until an asynchronous copy-out into ``a`` (``Vec.get`` under ``lazy_results``) has landed.  No-op for
    any other array.  Library calls that take ``a`` wait by themselves; NumPy code calls this first."""
    check(_lib.load().femo_host_wait(C.c_void_p(a.ctypes.data)))
    return a


def host_trim() -> None:
    """Release the recycled pinned blocks (femo_host_trim): between workloads of different sizes."""
    check(_lib.load().femo_host_trim())


def host_sync() -> None:
    """Wait for every asynchronous copy-out still in flight."""
    check(_lib.load().femo_host_sync())


# ---- caller arrays in pageable memory: pinned in place on second sight ---------------------------------------
# A backend that keeps its variables in NumPy arrays of its own hands the same arrays over in every operator call
# (utils_dolfinx.py:155-167, 300-311 copy from / into them).  The first transfer of such an array takes the staged
# path; when the same array (same owner object, address and size) shows up again it is pinned in place with
# femo_host_register, so every later transfer is one DMA at PCIe rate, and unpinned when NumPy releases the array
# (a weak-reference callback on the owning ndarray runs before its memory is freed).  No provenance is ever recorded
# for such memory (hostmem.cpp): its owner may write it at any time.  Arrays handed over once -- temporaries -- are
# never pinned: hipHostRegister of 0.5 GB costs more than ten staged transfers.
AUTO_REGISTER_MIN_BYTES = 1 << 20
_SEEN: Dict[int, tuple] = {}         # id(owner) -> (address, nbytes): seen once
_REGISTERED: Dict[int, tuple] = {}   # id(owner) -> (address, nbytes): pinned
_AUTO_REGISTER = True


def auto_register(enabled: bool) -> None:
    """Switch the pinning of repeatedly used caller arrays on / off (on by default)."""
    global _AUTO_REGISTER
    _AUTO_REGISTER = bool(enabled)


_REG_LOCK = threading.RLock()       # emulated ranks are threads of one process: the two tables and the register call are one critical section


def _forget_owner(key: int) -> None:
    with _REG_LOCK:
        _SEEN.pop(key, None)
        rec = _REGISTERED.pop(key, None)
        if rec is not None:
            try:
                _lib.load().femo_host_unregister(C.c_void_p(rec[0]))     # by the RECORDED address: the owner may have moved
            except Exception:
                pass


def _owner_of(a: np.ndarray):
    owner = a
    while isinstance(owner.base, np.ndarray):
        owner = owner.base
    if owner.base is not None or not owner.flags.owndata:
        return None                              # memory of another object (buffer, mmap, a pinned block): lifetime unknown / not ours
    return owner


def _note_caller_array(a: np.ndarray) -> None:
    """Called with every caller-owned array a transfer entry point is handed (``Vec.set`` / ``get(out=)`` / ``add_to_host``).
    ``ndarray.resize`` (an in-place realloc) of an array that was handed over is not supported: the old range would stay
    pinned until the array is seen again."""
    if not _AUTO_REGISTER or a.nbytes < AUTO_REGISTER_MIN_BYTES:
        return
    owner = _owner_of(a)
    if owner is None:
        return
    key, addr, nb = id(owner), owner.ctypes.data, owner.nbytes
    with _REG_LOCK:
        rec = _REGISTERED.get(key)
        if rec is not None:
            if rec == (addr, nb):
                return
            _forget_owner(key)                   # resized in place: unpin the old range (by its recorded address) first
        if _SEEN.get(key) != (addr, nb):
            if key not in _SEEN:
                import weakref
                weakref.finalize(owner, _forget_owner, key)
            _SEEN[key] = (addr, nb)
            return                               # first sight: staged path
        if _lib.load().femo_host_register(C.c_void_p(addr), nb) == 0:
            _REGISTERED[key] = (addr, nb)
        else:
            _SEEN.pop(key, None)                 # could not pin (limits, overlap): stay on the staged path


def register(a: np.ndarray) -> bool:
    """Pin a caller-owned array in place now (a driver that knows its variable storage up front need not wait for the
    second sight, and keeps hipHostRegister out of its first timed transfer).  Unpinned when NumPy frees the array.
    Returns False when the array cannot be pinned (not the owner of its memory, or the runtime refused)."""
    owner = _owner_of(a)
    if owner is None:
        return False
    key, addr, nb = id(owner), owner.ctypes.data, owner.nbytes
    with _REG_LOCK:
        rec = _REGISTERED.get(key)
        if rec == (addr, nb):
            return True
        if rec is not None:
            _forget_owner(key)
        if key not in _SEEN:
            import weakref
            weakref.finalize(owner, _forget_owner, key)
        _SEEN[key] = (addr, nb)
        if _lib.load().femo_host_register(C.c_void_p(addr), nb) == 0:
            _REGISTERED[key] = (addr, nb)
            return True
        return False


_LAZY = threading.local()


@contextlib.contextmanager
def lazy_results(enabled: bool = True):
    """Inside the block ``Vec.get()`` (without ``out``) returns its pinned array before the bytes have landed:
    the copy runs on the context's copy stream while later kernels execute.  The caller promises that whoever
    reads the array with code of its own calls ``host_wait`` / ``host_sync`` first (the operator classes hand
    such arrays only to a backend that set ``fea.async_results``)."""
    prev = getattr(_LAZY, "on", False)
    _LAZY.on = bool(enabled) or prev
    try:
        yield
    finally:
        _LAZY.on = prev


_DEFER = threading.local()


@contextlib.contextmanager
def deferred_uploads(enabled: bool = True):
    """Inside the block ``Vec.set`` from a pinned library block returns before the bytes have reached the device
    (femo_vec_set_host_deferred): the assembly entry points order their input-independent work in front of the wait, so
    the matrix of the first Newton pass, its scaling and the preconditioner weights run under the upload of the inputs
    (state_model.py:94-103).  On leaving the block the compute stream waits for every upload started in it, so code
    outside never sees a vector in flight."""
    if not enabled or getattr(_DEFER, "vecs", None) is not None:
        yield
        return
    _DEFER.vecs = []
    try:
        yield
    finally:
        vecs, _DEFER.vecs = _DEFER.vecs, None
        for v in vecs:
            v.await_upload()


def writable(a: np.ndarray, announce: bool = True) -> np.ndarray:
    """Writable alias of an array the library returned read-only.  For its new owner (a driver that
    accumulates into a result in place); the block stops counting as a mirror of its device vector.
    ``announce=False``: the alias will only be written by library calls (which keep the record straight
    themselves, and can use it: ``Vec.add_to_host`` into a block that still mirrors a device vector adds on the
    device); the owner calls ``host_touch`` before any write of its own."""
    if a.flags.writeable:
        return a
    base = a.base
    while isinstance(base, np.ndarray):                  # views of the returned array (ravel(), [:]) keep it as their base
        base = base.base
    if not isinstance(base, _PinnedBlock) or a.ctypes.data != base.ptr or a.size != base.n:
        raise ValueError("writable(): not a whole array returned by the engine")
    if announce:
        host_touch(a)
    return np.asarray(base)


def host_copy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[:] = src on the library's host threads (both contiguous fp64)."""
    assert dst.flags.c_contiguous and src.flags.c_contiguous and dst.dtype == src.dtype == np.float64 and dst.size == src.size
    check(_lib.load().femo_host_copy(C.c_void_p(dst.ctypes.data), C.c_void_p(src.ctypes.data), dst.size))


def host_axpby(a: float, x: np.ndarray, b: float, y: np.ndarray) -> None:
    """y = a x + b y on the library's host threads."""
    assert y.flags.c_contiguous and x.flags.c_contiguous and y.dtype == x.dtype == np.float64 and y.size == x.size
    check(_lib.load().femo_host_axpby(y.size, float(a), C.c_void_p(x.ctypes.data), float(b), C.c_void_p(y.ctypes.data)))


def host_stats(reset: bool = False) -> Dict[str, int]:
    st = _lib.HostStats()
    lib = _lib.load()
    check(lib.femo_host_get_stats(C.byref(st)))
    if reset:
        check(lib.femo_host_reset_stats())
    return {k: int(getattr(st, k)) for k, _ in _lib.HostStats._fields_}


class EmuGroup:
    """In-process rank emulation (tests): ``nranks`` contexts on one GPU, one host thread each, whose
    collectives go through host memory and barriers instead of RCCL (include/femo_hip.h)."""

    def __init__(self, nranks: int):
        self.lib = _lib.load()
        self.nranks = int(nranks)
        h = H()
        check(self.lib.femo_emu_group_create(self.nranks, C.byref(h)))
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.femo_emu_group_destroy(h)


def host_syncs(reset: bool = False) -> int:
    """Blocking host waits on the device the library executed since the last reset (femo_host_sync_stats)."""
    n = C.c_int64(0)
    check(_lib.load().femo_host_sync_stats(C.byref(n), int(reset)))
    return int(n.value)


class Context:
    """Device + HIP stream (+ RCCL communicator).  Stands in for PETSc/MPI global state."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        h = H()
        check(self.lib.femo_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = device
        self.rank, self.nranks = 0, 1

    def sync(self) -> None:
        check(self.lib.femo_ctx_sync(self.handle))

    @property
    def stream(self) -> int:
        return int(self.lib.femo_ctx_stream(self.handle) or 0)

    def comm_emulate(self, group: "EmuGroup", rank: int) -> None:
        check(self.lib.femo_comm_emulate(self.handle, group.handle, int(rank)))
        self.rank, self.nranks = int(rank), group.nranks
        self._emu_group = group            # keep it alive as long as the context

    def comm_model(self, rank: int, nranks: int) -> None:
        """Run the N-rank code paths alone on this GPU: collectives are counted but complete without moving data
        (`femo_comm_model`, include/femo_hip_test.h; bench.py's scaling model)."""
        check(self.lib.femo_comm_model(self.handle, int(rank), int(nranks)))
        self.rank, self.nranks = int(rank), int(nranks)

    def comm_init(self, unique_id: bytes, rank: int, nranks: int) -> None:
        assert len(unique_id) == 128
        check(self.lib.femo_comm_init(self.handle, unique_id, rank, nranks))
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().femo_comm_unique_id(buf))
        return buf.raw

    def comm_stats(self, reset: bool = False) -> Dict[str, int]:
        """Collectives this context issued since the last reset (femo_comm_stats)."""
        out = (C.c_int64 * 4)()
        check(self.lib.femo_comm_stats(self.handle, out, int(reset)))
        return dict(allreduce_calls=int(out[0]), allreduce_doubles=int(out[1]), neighbor_calls=int(out[2]), neighbor_doubles=int(out[3]))

    def allreduce_sum(self, values: Sequence[float]) -> np.ndarray:
        a = _f64(values).copy()
        check(self.lib.femo_allreduce_sum(self.handle, a.ctypes.data_as(_lib.c_f64p), a.size))
        return a

    def close(self) -> None:
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.femo_ctx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Vec:
    """fp64 device vector (PETSc Vec / dolfinx Function.vector)."""

    def __init__(self, ctx: Context, n: int, device_ptr: Optional[int] = None):
        self.ctx, self.lib, self.n = ctx, ctx.lib, int(n)
        h = H()
        if device_ptr is None:
            check(self.lib.femo_vec_create(ctx.handle, self.n, C.byref(h)))
        else:
            check(self.lib.femo_vec_wrap(ctx.handle, C.c_void_p(device_ptr), self.n, C.byref(h)))
        self.handle = h

    def set(self, a) -> "Vec":
        a = _f64(a)
        _note_caller_array(a)
        pending = getattr(_DEFER, "vecs", None)
        if pending is not None:
            # inside engine.deferred_uploads(): the copy may still be in flight when this returns (pinned library blocks only)
            check(self.lib.femo_vec_set_host_deferred(self.handle, _ptr(a), a.size))
            pending.append(self)
        else:
            check(self.lib.femo_vec_set_host(self.handle, _ptr(a), a.size))
        return self

    def await_upload(self) -> "Vec":
        check(self.lib.femo_vec_await_upload(self.handle))
        return self

    def get(self, n: Optional[int] = None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Host copy.  Without ``out`` the result lives in a pinned block (one DMA, no page faults of a
        fresh allocation) and is READ-ONLY: the library remembers it as an exact copy of this vector,
        so handing it back to ``set`` moves no PCIe bytes while the vector is unchanged
        (include/femo_hip.h, "host memory").  ``engine.writable(a)`` gives its owner a writable alias."""
        n = self.n if n is None else int(n)
        if out is not None:
            assert out.dtype == np.float64 and out.flags.c_contiguous and out.size == n
            _note_caller_array(out)
            check(self.lib.femo_vec_get_host(self.handle, _ptr(out), n))
            return out
        out = pinned_empty(n)
        if getattr(_LAZY, "on", False):
            check(self.lib.femo_vec_get_host_async(self.handle, _ptr(out), n))
        else:
            check(self.lib.femo_vec_get_host(self.handle, _ptr(out), n))
        out.flags.writeable = False
        return out

    def add_to_host(self, host: np.ndarray, n: Optional[int] = None) -> np.ndarray:
        """host[:n] += self[:n] in place (contiguous fp64 array)."""
        n = self.n if n is None else int(n)
        assert host.dtype == np.float64 and host.flags.c_contiguous and host.flags.writeable and host.size >= n
        _note_caller_array(host)
        check(self.lib.femo_vec_add_to_host(self.handle, _ptr(host), n))
        return host

    def fill(self, value: float) -> "Vec":
        check(self.lib.femo_vec_fill(self.handle, float(value)))
        return self

    def copy_from(self, other: "Vec") -> "Vec":
        check(self.lib.femo_vec_copy(self.handle, other.handle))
        return self

    def axpy(self, a: float, x: "Vec") -> "Vec":
        check(self.lib.femo_vec_axpy(self.handle, float(a), x.handle))
        return self

    def dot(self, other: "Vec", n: Optional[int] = None) -> float:
        out = C.c_double(0.0)
        check(self.lib.femo_vec_dot(self.handle, other.handle, self.n if n is None else n, C.byref(out)))
        return out.value

    @staticmethod
    def dots(pairs, n: int) -> List[float]:
        """[x . y for (x, y) in pairs] over the first ``n`` entries, at most four pairs: one kernel, one reduction
        and one host synchronisation for all of them (femo_vec_dots)."""
        k = len(pairs)
        xs = (C.c_void_p * k)(*[p[0].handle for p in pairs])
        ys = (C.c_void_p * k)(*[p[1].handle for p in pairs])
        out = (C.c_double * k)()
        check(pairs[0][0].lib.femo_vec_dots(k, xs, ys, int(n), out))
        return [float(v) for v in out]

    @staticmethod
    def dots_rhs(pairs, n: int, mat: "Mat", rhs: "Vec") -> List[float]:
        """``dots(pairs, n)`` plus, as the last value, rho_0 = sum over the non-identity rows of (rhs_i / sqrt(diag_i))^2 for
        the operator ``mat``: what a Krylov solve of it with right-hand side ``rhs`` starts from (femo_vec_dots_rhs) -- the
        same launch, reduction and host synchronisation."""
        k = len(pairs)
        xs = (C.c_void_p * k)(*[p[0].handle for p in pairs])
        ys = (C.c_void_p * k)(*[p[1].handle for p in pairs])
        out = (C.c_double * (k + 1))()
        check(pairs[0][0].lib.femo_vec_dots_rhs(k, xs, ys, int(n), out, mat.handle, rhs.handle))
        return [float(v) for v in out]

    @property
    def device_ptr(self) -> int:
        """Mutable device address: the library assumes the holder writes through it (include/femo_hip.h)."""
        return int(self.lib.femo_vec_device_ptr(self.handle) or 0)

    @property
    def device_ptr_const(self) -> int:
        return int(self.lib.femo_vec_device_ptr_const(self.handle) or 0)

    def __del__(self):
        # destroy only while the owning context is alive (interpreter shutdown tears
        # objects down in arbitrary order; a dead parent means the device is gone too)
        try:
            h, self.handle = getattr(self, "handle", None), None
            if h and getattr(self.ctx, "handle", None):
                self.lib.femo_vec_destroy(h)
        except Exception:
            pass


class DeviceArray:
    """A value that stays in HBM while it crosses the operator boundary.

    The CSDL backend hands NumPy arrays to the operators; the in-repo driver
    (and bench.py) may hand these instead, so that ``update`` becomes a
    device-to-device copy and ``getFuncArray`` no PCIe transfer.  Supports the
    three things the operators do with a value: ``len``, ``+=`` and conversion
    to NumPy."""

    __array_priority__ = 100

    def __init__(self, vec: Vec, n: Optional[int] = None):
        self.vec = vec
        self.n = vec.n if n is None else int(n)

    def __len__(self) -> int:
        return self.n

    @property
    def shape(self):
        return (self.n,)

    def __iadd__(self, other):
        if isinstance(other, DeviceArray):
            self.vec.axpy(1.0, other.vec)
        else:
            tmp = Vec(self.vec.ctx, self.vec.n).set(np.asarray(other, dtype=np.float64))
            self.vec.axpy(1.0, tmp)
        return self

    def numpy(self) -> np.ndarray:
        return self.vec.get(self.n)

    def __array__(self, dtype=None, copy=None):
        return self.numpy()

    def copy(self) -> "DeviceArray":
        return DeviceArray(Vec(self.vec.ctx, self.vec.n).copy_from(self.vec), self.n)

    @staticmethod
    def zeros(ctx: Context, n: int) -> "DeviceArray":
        return DeviceArray(Vec(ctx, n), n)


class DeviceMesh:
    """P1 simplex mesh on the device + incidence + sparsity pattern (dolfinx Mesh + dofmap)."""

    def __init__(self, ctx: Context, x: np.ndarray, conn: np.ndarray, n_rows: Optional[int] = None):
        self.ctx, self.lib = ctx, ctx.lib
        x = _f64(x)
        conn = _i32(conn)
        self.tdim = x.shape[1]
        assert conn.shape[1] == self.tdim + 1
        self.n_vert, self.n_cell = x.shape[0], conn.shape[0]
        self.n_rows = self.n_vert if n_rows is None else int(n_rows)
        h = H()
        check(self.lib.femo_mesh_create(ctx.handle, self.tdim, self.n_vert, self.n_rows, _ptr(x),
                                        self.n_cell, _ptr(conn), C.byref(h)))
        self.handle = h
        self.info = self._info()

    def _info(self) -> Dict[str, int]:
        buf = (C.c_int64 * _lib.MESH_INFO_COUNT)()
        check(self.lib.femo_mesh_info(self.handle, buf))
        return dict(zip(_lib.MESH_INFO_KEYS, (int(v) for v in buf)))

    def pattern_csr(self):
        rowptr = np.zeros(self.n_rows + 1, np.int64)
        col = np.zeros(self.info["nnz"], np.int32)
        check(self.lib.femo_mesh_pattern_csr(self.handle, _ptr(rowptr), _ptr(col)))
        return rowptr, col

    def set_boundary_facets(self, mask: Optional[np.ndarray]) -> None:
        """mask[c] bit k <=> facet of cell c opposite local vertex k is an exterior facet."""
        if mask is None:
            check(self.lib.femo_mesh_set_boundary_facets(self.handle, None))
            return
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert mask.shape == (self.n_cell,)
        check(self.lib.femo_mesh_set_boundary_facets(self.handle, _ptr(mask)))

    def set_global(self, lo, hi, n_vert_global: int) -> None:
        """Whole-mesh bounding box / vertex count of a partitioned mesh (BPX lattice geometry)."""
        lo, hi = np.ascontiguousarray(lo, np.float64), np.ascontiguousarray(hi, np.float64)
        check(self.lib.femo_mesh_set_global(self.handle, _ptr(lo), _ptr(hi), int(n_vert_global)))

    def pc_info(self):
        nl, nodes = C.c_int32(0), C.c_int64(0)
        check(self.lib.femo_mesh_pc_info(self.handle, C.byref(nl), C.byref(nodes)))
        return {"levels": nl.value, "finest_nodes": nodes.value}

    def set_halo(self, nbr, send_ptr, send_idx, recv_ptr) -> None:
        nbr = _i32(nbr)
        send_ptr = np.ascontiguousarray(send_ptr, np.int64)
        recv_ptr = np.ascontiguousarray(recv_ptr, np.int64)
        send_idx = _i32(send_idx)
        check(self.lib.femo_mesh_set_halo(self.handle, len(nbr), _ptr(nbr), _ptr(send_ptr),
                                          _ptr(send_idx), _ptr(recv_ptr)))

    def halo_exchange(self, v: Vec) -> None:
        check(self.lib.femo_halo_exchange(self.handle, v.handle))

    # -- device-initiated ghost refresh (include/femo_hip.h, ABI 9): the collective set-up lives in femo_amd/dist --
    def halo_direct_export(self):
        """(64-byte IPC handle, device address, workgroups per producer launch) of this rank's inbox."""
        buf = C.create_string_buffer(64)
        addr, nb = C.c_uint64(0), C.c_int32(0)
        check(self.lib.femo_mesh_halo_direct_export(self.handle, buf, C.byref(addr), C.byref(nb)))
        return buf.raw, int(addr.value), int(nb.value)

    def halo_direct_connect(self, mode: int, handles, addresses, remote_offset, remote_n_ghost, remote_slot, remote_blocks) -> None:
        hb = b"".join(handles) if handles else None
        addresses = np.ascontiguousarray(addresses, np.uint64)
        ro, rg = np.ascontiguousarray(remote_offset, np.int64), np.ascontiguousarray(remote_n_ghost, np.int64)
        rs, rb = _i32(remote_slot), _i32(remote_blocks)
        check(self.lib.femo_mesh_halo_direct_connect(self.handle, int(mode), hb, _ptr(addresses), _ptr(ro), _ptr(rg), _ptr(rs), _ptr(rb)))

    def halo_direct_selftest(self) -> bool:
        ok = C.c_int(0)
        check(self.lib.femo_mesh_halo_direct_selftest(self.handle, C.byref(ok)))
        return bool(ok.value)

    def halo_direct_enable(self, on: bool) -> None:
        check(self.lib.femo_mesh_halo_direct_enable(self.handle, int(bool(on))))

    def halo_direct_info(self) -> Dict[str, int]:
        out = (C.c_int64 * 4)()
        check(self.lib.femo_mesh_halo_direct_info(self.handle, out))
        return dict(enabled=int(out[0]), exchanges=int(out[1]), timeouts=int(out[2]), producer_blocks=int(out[3]))

    def __del__(self):
        try:
            h, self.handle = getattr(self, "handle", None), None
            if h and getattr(self.ctx, "handle", None):
                self.lib.femo_mesh_destroy(h)
        except Exception:
            pass


class DirichletSet:
    """Strong Dirichlet dofs + values (list of dolfinx dirichletbc)."""

    def __init__(self, mesh: DeviceMesh, dofs, vals):
        self.mesh, self.lib = mesh, mesh.lib
        dofs = _i32(dofs)
        vals = _f64(np.broadcast_to(vals, dofs.shape))
        self.dofs, self.vals = dofs, vals
        h = H()
        check(self.lib.femo_bc_create(mesh.handle, dofs.size, _ptr(dofs), _ptr(vals), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            h, self.handle = getattr(self, "handle", None), None
            if h and getattr(self.mesh, "handle", None) and getattr(self.mesh.ctx, "handle", None):
                self.lib.femo_bc_destroy(h)
        except Exception:
            pass


class Mat:
    """N x N sparse matrix on the mesh pattern (PETSc Mat)."""

    def __init__(self, mesh: DeviceMesh):
        self.mesh, self.lib = mesh, mesh.lib
        h = H()
        check(self.lib.femo_mat_create(mesh.handle, C.byref(h)))
        self.handle = h

    def mult(self, x: Vec, y: Vec, transpose: bool = False) -> Vec:
        check(self.lib.femo_mat_spmv(self.handle, int(transpose), x.handle, y.handle))
        return y

    def identity_solve(self, b: Vec, x: Vec) -> Vec:
        """x = the result of a solve that decides not to iterate from the zero guess: b_i / diag_i on the identity rows of the
        last assembly, 0 elsewhere (femo_mat_identity_solve)."""
        check(self.lib.femo_mat_identity_solve(self.handle, b.handle, x.handle))
        return x

    def prescale(self) -> "Mat":
        """S = diag^-1/2 and S A S now (`femo_mat_prescale`): what the first solve with this matrix would form first."""
        check(self.lib.femo_mat_prescale(self.handle))
        return self

    def export_csr(self):
        m = self.mesh
        rowptr = np.zeros(m.n_rows + 1, np.int64)
        col = np.zeros(m.info["nnz"], np.int32)
        val = np.zeros(m.info["nnz"], np.float64)
        check(self.lib.femo_mat_export_csr(self.handle, _ptr(rowptr), _ptr(col), _ptr(val)))
        return rowptr, col, val

    def to_scipy(self):
        import scipy.sparse as sp
        rowptr, col, val = self.export_csr()
        return sp.csr_matrix((val, col, rowptr), shape=(self.mesh.n_rows, self.mesh.n_vert))

    def diagonal(self, out: Vec) -> Vec:
        check(self.lib.femo_mat_diagonal(self.handle, out.handle))
        return out

    def solve_cg(self, b: Vec, x: Vec, transpose: bool = False, rtol: float = 1e-12, atol: float = 0.0,
                 max_it: int = 100000, zero_guess: bool = True, check_every: int = 32, pc: str = "jacobi",
                 atol_pc: float = 0.0) -> SolveInfo:
        """Stopping rules: include/femo_hip.h, femo_solver_opts (rtol acts in the Jacobi norm for pc='jacobi',
        in the norm of the preconditioner for pc='bpx')."""
        opts = SolverOpts(rtol, atol, max_it, int(zero_guess), check_every, PC_KINDS[pc], atol_pc)
        info = SolveInfo()
        check(self.lib.femo_solve_cg(self.handle, int(transpose), b.handle, x.handle, C.byref(opts), C.byref(info)))
        return info

    def pc_apply(self, r: Vec, z: Vec) -> Vec:
        """z = M^-1 r with the BPX preconditioner of this operator (unscaled variables)."""
        check(self.lib.femo_mat_pc_apply(self.handle, r.handle, z.handle))
        return z

    def solve_bicgstab(self, b: Vec, x: Vec, transpose: bool = False, rtol: float = 1e-12, atol: float = 0.0,
                       max_it: int = 100000, zero_guess: bool = True, check_every: int = 32) -> SolveInfo:
        opts = SolverOpts(rtol, atol, max_it, int(zero_guess), check_every, 0, 0.0)
        info = SolveInfo()
        check(self.lib.femo_solve_bicgstab(self.handle, int(transpose), b.handle, x.handle, C.byref(opts), C.byref(info)))
        return info

    def bench_spmv(self, x: Vec, y: Vec, reps: int = 50) -> float:
        ms = C.c_double(0.0)
        check(self.lib.femo_bench_spmv(self.handle, x.handle, y.handle, reps, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            h, self.handle = getattr(self, "handle", None), None
            if h and getattr(self.mesh, "handle", None) and getattr(self.mesh.ctx, "handle", None):
                self.lib.femo_mat_destroy(h)
        except Exception:
            pass


def _params(params) -> Optional[np.ndarray]:
    if params is None:
        return None
    p = np.zeros(8)
    p[:len(params)] = params
    return p


def assemble_residual(mesh: DeviceMesh, pde: int, params, u: Vec, f: Vec, r: Vec, aux: Optional[Vec] = None) -> Vec:
    check(mesh.lib.femo_assemble_residual(mesh.handle, pde, _ptr(_params(params)), u.handle, f.handle, _h(aux), r.handle))
    return r


def assemble_jacobian(mesh: DeviceMesh, pde: int, params, u: Optional[Vec], f: Optional[Vec],
                      bc: Optional[DirichletSet], J: Mat, aux: Optional[Vec] = None) -> Mat:
    check(mesh.lib.femo_assemble_jacobian(mesh.handle, pde, _ptr(_params(params)), _h(u), _h(f), _h(aux), _h(bc), J.handle))
    return J


def assemble_system(mesh: DeviceMesh, pde: int, params, u: Optional[Vec], f: Optional[Vec],
                    bc: Optional[DirichletSet], J_nobc: Optional[Mat], A_bc: Optional[Mat],
                    rhs: Optional[Vec], aux: Optional[Vec] = None) -> None:
    """One pass for any subset of dR/du (no BCs), A (BCs eliminated) and the Newton rhs."""
    check(mesh.lib.femo_assemble_system(mesh.handle, pde, _ptr(_params(params)), _h(u), _h(f), _h(aux), _h(bc),
                                        _h(J_nobc), _h(A_bc), _h(rhs)))


def bc_apply_rhs(bc: DirichletSet, u: Vec, b: Vec) -> Vec:
    check(bc.lib.femo_bc_apply_rhs(bc.handle, u.handle, b.handle))
    return b


def assemble_dRdf(mesh: DeviceMesh, pde: int, params, u: Optional[Vec], f: Optional[Vec], vals: Vec) -> Vec:
    check(mesh.lib.femo_assemble_dRdf(mesh.handle, pde, _ptr(_params(params)), _h(u), _h(f), vals.handle))
    return vals


def assemble_dRdf_cell(mesh: DeviceMesh, pde: int, params, cvals: Vec) -> Vec:
    """dR/df of a Poisson-type form in its compact shape: one value per cell (include/femo_hip.h)."""
    check(mesh.lib.femo_assemble_dRdf_cell(mesh.handle, pde, _ptr(_params(params)), cvals.handle))
    return cvals


def dRdf_cell_apply(mesh: DeviceMesh, cvals: Vec, x: Vec, y: Vec, transpose: bool, accumulate: bool = False) -> Vec:
    check(mesh.lib.femo_dRdf_cell_apply(mesh.handle, cvals.handle, int(transpose), x.handle, y.handle, int(accumulate)))
    return y


def newton_rhs(K: Mat, F: Vec, u: Vec, bc: DirichletSet, b: Vec) -> Vec:
    check(K.lib.femo_newton_rhs(K.handle, F.handle, u.handle, bc.handle, b.handle))
    return b


def dRdf_apply(mesh: DeviceMesh, vals: Vec, x: Vec, y: Vec, transpose: bool, accumulate: bool = False) -> Vec:
    check(mesh.lib.femo_dRdf_apply(mesh.handle, vals.handle, int(transpose), x.handle, y.handle, int(accumulate)))
    return y


def functional_value(mesh: DeviceMesh, kind: int, params, u: Vec, f: Vec, u_d: Vec) -> float:
    out = C.c_double(0.0)
    check(mesh.lib.femo_functional_value(mesh.handle, kind, _ptr(_params(params)), u.handle, f.handle, u_d.handle, C.byref(out)))
    return out.value


def functional_grad_u(mesh: DeviceMesh, kind: int, params, u: Vec, f: Optional[Vec], u_d: Vec, g: Vec) -> Vec:
    check(mesh.lib.femo_functional_grad_u(mesh.handle, kind, _ptr(_params(params)), u.handle, _h(f), u_d.handle, g.handle))
    return g


def functional_grad_f(mesh: DeviceMesh, kind: int, params, u: Optional[Vec], f: Vec, u_d: Optional[Vec], g: Vec) -> Vec:
    check(mesh.lib.femo_functional_grad_f(mesh.handle, kind, _ptr(_params(params)), _h(u), f.handle, _h(u_d), g.handle))
    return g


def cell_expression(mesh: DeviceMesh, kind: int, params, vin: Vec, out: Vec) -> Vec:
    check(mesh.lib.femo_cell_expression(mesh.handle, kind, _ptr(_params(params)), vin.handle, out.handle))
    return out


def pointwise_divide(y: Vec, x: Vec, d: Vec, n: int) -> Vec:
    check(y.lib.femo_vec_pointwise_divide(y.handle, x.handle, d.handle, n))
    return y


def topology_host(tdim: int, n_vert: int, n_rows: int, conn: np.ndarray):
    """Host-only pattern build (no GPU): returns (info dict, rowptr, col)."""
    lib = _lib.load()
    conn = _i32(conn)
    info = (C.c_int64 * _lib.MESH_INFO_COUNT)()
    check(lib.femo_topology_build_host(tdim, n_vert, n_rows, conn.shape[0], _ptr(conn), info, None, None))
    d = dict(zip(_lib.MESH_INFO_KEYS, (int(v) for v in info)))
    rowptr = np.zeros(n_rows + 1, np.int64)
    col = np.zeros(d["nnz"], np.int32)
    check(lib.femo_topology_build_host(tdim, n_vert, n_rows, conn.shape[0], _ptr(conn), info, _ptr(rowptr), _ptr(col)))
    return d, rowptr, col


def pc_plan_host(x: np.ndarray, lo=None, hi=None, n_vert_global: Optional[int] = None) -> Dict:
    """Host-only plan of the BPX lattice for the vertices ``x`` (no GPU): levels, bins, packed lattice
    coordinates and the (brick, bin) sort the restriction kernel walks."""
    lib = _lib.load()
    x = _f64(x)
    n, d = x.shape
    lo = _f64(x.min(axis=0) if lo is None else lo)
    hi = _f64(x.max(axis=0) if hi is None else hi)
    ng = int(n_vert_global or n)
    nl, nb = C.c_int32(0), C.c_int64(0)
    check(lib.femo_pc_plan_host(d, n, _ptr(x), _ptr(lo), _ptr(hi), ng, C.byref(nl), None, C.byref(nb),
                                None, None, None, None, None))
    bins = np.zeros((nl.value, 3), np.int32)
    pk = np.zeros((n, 2), np.uint32)          # 8 bytes per vertex in both dimensions (femo_internal.h: FEMO_PK_WORDS)
    perm = np.zeros(n, np.int32)
    brick_ptr = np.zeros(nb.value + 1, np.int64)
    brick_base = np.zeros((max(nb.value, 1), 3), np.int32)
    bin_ptr = np.zeros((max(nb.value, 1), 65), np.uint32)
    check(lib.femo_pc_plan_host(d, n, _ptr(x), _ptr(lo), _ptr(hi), ng, C.byref(nl), _ptr(bins), C.byref(nb),
                                _ptr(pk), _ptr(perm), _ptr(brick_ptr), _ptr(brick_base), _ptr(bin_ptr)))
    counts = np.diff(bin_ptr[:nb.value].astype(np.int64), axis=1).ravel()
    filled = counts[counts > 0]
    occupancy = float(filled.max() / filled.mean()) if filled.size else 1.0
    return dict(levels=nl.value, bins=bins[:, :d], n_bricks=nb.value, pk=pk, perm=perm, brick_ptr=brick_ptr,
                brick_base=brick_base[:nb.value], bin_ptr=bin_ptr[:nb.value], occupancy=occupancy)
