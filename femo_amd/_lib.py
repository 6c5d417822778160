"""ctypes binding of ``libfemo_hip.so`` (include/femo_hip.h).

The product has no CPU fallback: if the shared library is missing, or a compute
entry point is called without a HIP device, an exception is raised -- nothing is
silently routed elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEMO_LIB") or os.path.join(_HERE, "csrc", "libfemo_hip.so")    # FEMO_LIB: another build of the library (A/B measurements)

c_i64 = C.c_int64
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
H = C.c_void_p  # opaque handle

ABI_VERSION = 10            # include/femo_hip.h FEMO_ABI_VERSION
MESH_INFO_COUNT = 12
MESH_INFO_KEYS = ("tdim", "n_vert", "n_rows", "n_cell", "nnz", "sell_entries", "max_rowlen",
                  "max_valence", "n_slices", "visit_entries", "regular_slices", "short_slices")

PDE_POISSON = 0
PDE_NL_POISSON = 1
PDE_MASS = 2
PDE_EB_BEAM = 3
J_L2_TRACKING = 0


class SolverOpts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("max_it", C.c_int32),
                ("zero_guess", C.c_int32), ("check_every", C.c_int32), ("pc", C.c_int32), ("atol_pc", C.c_double)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("residual_norm", C.c_double), ("pc_residual_norm", C.c_double), ("pc_rhs_norm", C.c_double),
                ("rhs_norm", C.c_double),
                ("solve_ms", C.c_double), ("spmv_ms", C.c_double),
                ("spmv_samples", C.c_int32), ("loop_allreduces", C.c_int32)]


class HostStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("h2d_pinned", "h2d_pinned_bytes", "h2d_staged", "h2d_staged_bytes",
                                         "h2d_skipped", "h2d_skipped_bytes", "h2d_as_d2d", "h2d_as_d2d_bytes",
                                         "d2h_pinned", "d2h_pinned_bytes", "d2h_staged", "d2h_staged_bytes",
                                         "d2h_async", "d2h_async_bytes", "d2h_device_sum", "d2h_device_sum_bytes",
                                         "h2d_deferred", "h2d_deferred_bytes")]


# name -> (restype, argtypes); every symbol include/femo_hip.h declares
PROTOTYPES = {
    "femo_last_error": (C.c_char_p, []),
    "femo_abi_version": (C.c_int, []),
    "femo_host_sync_stats": (C.c_int, [C.POINTER(C.c_int64), C.c_int]),
    "femo_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "femo_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(H)]),
    "femo_ctx_destroy": (C.c_int, [H]),
    "femo_ctx_sync": (C.c_int, [H]),
    "femo_ctx_stream": (C.c_void_p, [H]),
    "femo_vec_create": (C.c_int, [H, c_i64, C.POINTER(H)]),
    "femo_vec_wrap": (C.c_int, [H, C.c_void_p, c_i64, C.POINTER(H)]),
    "femo_vec_destroy": (C.c_int, [H]),
    "femo_vec_size": (c_i64, [H]),
    "femo_vec_device_ptr": (C.c_void_p, [H]),
    "femo_vec_device_ptr_const": (C.c_void_p, [H]),
    "femo_vec_set_host": (C.c_int, [H, C.c_void_p, c_i64]),
    "femo_vec_get_host": (C.c_int, [H, C.c_void_p, c_i64]),
    "femo_vec_add_to_host": (C.c_int, [H, C.c_void_p, c_i64]),
    "femo_vec_get_host_async": (C.c_int, [H, C.c_void_p, c_i64]),
    "femo_vec_set_host_deferred": (C.c_int, [H, C.c_void_p, c_i64]),
    "femo_vec_await_upload": (C.c_int, [H]),
    "femo_host_wait": (C.c_int, [C.c_void_p]),
    "femo_host_sync": (C.c_int, []),
    "femo_host_alloc": (C.c_int, [c_i64, C.POINTER(C.c_void_p)]),
    "femo_host_free": (C.c_int, [C.c_void_p]),
    "femo_host_trim": (C.c_int, []),
    "femo_host_register": (C.c_int, [C.c_void_p, c_i64]),
    "femo_host_unregister": (C.c_int, [C.c_void_p]),
    "femo_host_touch": (C.c_int, [C.c_void_p]),
    "femo_host_fill": (C.c_int, [C.c_void_p, c_i64, C.c_double]),
    "femo_host_is_pinned": (C.c_int, [C.c_void_p, c_i64]),
    "femo_host_threads": (C.c_int, []),
    "femo_host_copy": (C.c_int, [C.c_void_p, C.c_void_p, c_i64]),
    "femo_host_axpby": (C.c_int, [c_i64, C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "femo_host_get_stats": (C.c_int, [C.POINTER(HostStats)]),
    "femo_host_reset_stats": (C.c_int, []),
    "femo_vec_fill": (C.c_int, [H, C.c_double]),
    "femo_vec_copy": (C.c_int, [H, H]),
    "femo_vec_axpy": (C.c_int, [H, C.c_double, H]),
    "femo_vec_dot": (C.c_int, [H, H, c_i64, c_f64p]),
    "femo_vec_dots": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_i64, c_f64p]),
    "femo_vec_dots_rhs": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_i64, c_f64p, H, H]),
    "femo_mat_identity_solve": (C.c_int, [H, H, H]),
    "femo_mesh_create": (C.c_int, [H, C.c_int, c_i64, c_i64, C.c_void_p, c_i64, C.c_void_p, C.POINTER(H)]),
    "femo_mesh_destroy": (C.c_int, [H]),
    "femo_mesh_info": (C.c_int, [H, c_i64p]),
    "femo_mesh_set_boundary_facets": (C.c_int, [H, C.c_void_p]),
    "femo_pc_plan_host": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32),
                          C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_emu_group_create": (C.c_int, [C.c_int, C.POINTER(H)]),
    "femo_emu_group_destroy": (C.c_int, [H]),
    "femo_comm_emulate": (C.c_int, [H, H, C.c_int]),
    "femo_comm_model": (C.c_int, [H, C.c_int, C.c_int]),
    "femo_comm_stats": (C.c_int, [H, C.POINTER(C.c_int64), C.c_int]),
    # Reissner-Mindlin shell
    "femo_shell_create": (C.c_int, [H, c_i64, C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(H)]),
    "femo_shell_destroy": (C.c_int, [H]),
    "femo_shell_pc_create": (C.c_int, [H, C.c_int, c_i64, C.c_int, C.c_void_p] + [C.c_void_p] * 11),
    "femo_shell_pc_coarse": (C.c_int, [H, C.c_int, C.c_void_p, c_i64] + [C.c_void_p] * 6),
    "femo_shell_pc_coarse_matrix": (C.c_int, [H, H, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(c_i64)]),
    "femo_shell_pc_hermite": (C.c_int, [H] + [C.c_void_p] * 11),
    "femo_shell_pc_info": (C.c_int, [H, C.POINTER(C.c_int32)]),
    "femo_shell_pc_block_items": (C.c_int, [H, C.c_int64] + [C.c_void_p] * 4),
    "femo_shell_pc_weights": (C.c_int, [H, C.c_double, C.c_double]),
    "femo_shell_pc_apply": (C.c_int, [H, H, C.c_void_p, H, H]),
    "femo_shell_pnorm_stress": (C.c_int, [H, C.c_double, C.c_double, H, H, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.POINTER(C.c_double), C.c_int, H, H]),
    "femo_shell_vm_rhs": (C.c_int, [H, C.c_double, C.c_double, H, H, C.c_double, H, H]),
    "femo_shell_p1_mass": (C.c_int, [H, H, H]),
    "femo_shell_ndof": (c_i64, [H]),
    "femo_shell_nnz": (c_i64, [H]),
    "femo_shell_assemble": (C.c_int, [H, C.c_double, C.c_double, H, H]),
    "femo_shell_matvec": (C.c_int, [H, H, C.c_void_p, H, H]),
    "femo_shell_load": (C.c_int, [H, H, C.c_double, C.c_int, H]),
    "femo_shell_load_T": (C.c_int, [H, H, C.c_double, C.c_int, H]),
    "femo_shell_dform_dh": (C.c_int, [H, C.c_double, C.c_double, H, H, H, C.c_int, H, C.POINTER(C.c_double)]),
    "femo_shell_compliance": (C.c_int, [H, H, C.POINTER(C.c_double), C.c_int, H]),
    "femo_shell_mass": (C.c_int, [H, C.c_double, H, C.POINTER(C.c_double), C.c_int, H]),
    "femo_shell_compliance_dx": (C.c_int, [H, H, H, C.POINTER(C.c_double), C.c_int, H]),
    "femo_shell_regularization": (C.c_int, [H, C.c_int, H, C.POINTER(C.c_double), C.c_int, H]),
    "femo_shell_hpower": (C.c_int, [H, C.c_double, C.c_double, H, C.POINTER(C.c_double), C.c_int, H]),
    "femo_shell_set_penalty": (C.c_int, [H, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_shell_penalty_add": (C.c_int, [H, H]),
    "femo_shell_penalty_apply": (C.c_int, [H, H, H, C.c_int, H]),
    "femo_shell_inertia_apply": (C.c_int, [H, C.c_double, H, H, C.c_int, H]),
    "femo_shell_inertia_dh": (C.c_int, [H, C.c_double, H, H, H, C.c_int, H]),
    "femo_shell_inertia_dh_fwd": (C.c_int, [H, C.c_double, H, H, H, C.c_int, H]),
    "femo_shell_dform_dh_fwd": (C.c_int, [H, C.c_double, C.c_double, H, H, H, C.c_int, H]),
    "femo_shell_solve": (C.c_int, [H, H, C.c_void_p, H, H, H, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]),
    "femo_shell_set_partition": (C.c_int, [H, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_shell_set_owned_cells": (C.c_int, [H, C.c_void_p]),
    "femo_shell_halo": (C.c_int, [H, H]),
    "femo_shell_mask_unowned": (C.c_int, [H, H]),
    "femo_mesh_set_global": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int64]),
    "femo_mesh_pc_info": (C.c_int, [H, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "femo_mesh_pattern_csr": (C.c_int, [H, C.c_void_p, C.c_void_p]),
    "femo_topology_build_host": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, C.c_void_p, c_i64p, C.c_void_p, C.c_void_p]),
    "femo_bc_create": (C.c_int, [H, c_i64, C.c_void_p, C.c_void_p, C.POINTER(H)]),
    "femo_bc_destroy": (C.c_int, [H]),
    "femo_assemble_residual": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, H]),
    "femo_mat_create": (C.c_int, [H, C.POINTER(H)]),
    "femo_mat_destroy": (C.c_int, [H]),
    "femo_assemble_jacobian": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, H, H]),
    "femo_assemble_dRdf": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H]),
    "femo_assemble_system": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, H, H, H, H]),
    "femo_bc_apply_rhs": (C.c_int, [H, H, H]),
    "femo_newton_rhs": (C.c_int, [H, H, H, H, H]),
    "femo_mat_spmv": (C.c_int, [H, C.c_int, H, H]),
    "femo_dRdf_apply": (C.c_int, [H, H, C.c_int, H, H, C.c_int]),
    "femo_assemble_dRdf_cell": (C.c_int, [H, C.c_int, C.c_void_p, H]),
    "femo_dRdf_cell_apply": (C.c_int, [H, H, C.c_int, H, H, C.c_int]),
    "femo_mat_export_csr": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_mat_diagonal": (C.c_int, [H, H]),
    "femo_mat_prescale": (C.c_int, [H]),
    "femo_solve_cg": (C.c_int, [H, C.c_int, H, H, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]),
    "femo_mat_pc_apply": (C.c_int, [H, H, H]),
    "femo_solve_bicgstab": (C.c_int, [H, C.c_int, H, H, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]),
    "femo_functional_value": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, c_f64p]),
    "femo_functional_grad_u": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, H]),
    "femo_functional_grad_f": (C.c_int, [H, C.c_int, C.c_void_p, H, H, H, H]),
    "femo_cell_expression": (C.c_int, [H, C.c_int, C.c_void_p, H, H]),
    "femo_vec_pointwise_divide": (C.c_int, [H, H, H, c_i64]),
    "femo_bench_spmv": (C.c_int, [H, H, H, C.c_int, c_f64p]),
    "femo_comm_unique_id": (C.c_int, [C.c_char_p]),
    "femo_comm_init": (C.c_int, [H, C.c_char_p, C.c_int, C.c_int]),
    "femo_comm_rank": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "femo_mesh_set_halo": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_halo_exchange": (C.c_int, [H, H]),
    "femo_mesh_halo_direct_export": (C.c_int, [H, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "femo_mesh_halo_direct_connect": (C.c_int, [H, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "femo_mesh_halo_direct_selftest": (C.c_int, [H, C.POINTER(C.c_int)]),
    "femo_mesh_halo_direct_enable": (C.c_int, [H, C.c_int]),
    "femo_mesh_halo_direct_info": (C.c_int, [H, C.POINTER(C.c_int64)]),
    "femo_allreduce_sum": (C.c_int, [H, c_f64p, C.c_int]),
}


class FemoError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libfemo_hip.so and bind every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FemoError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C femo_amd/csrc`).  femo_amd has no CPU fallback.")
    # A context drives three HIP streams (compute, copies, neighbour exchange) and RCCL adds its own.  The HIP runtime maps
    # streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): streams that SHARE a queue serialise -- a copy stream's wait
    # for a 60 MB PCIe transfer then holds up the compute stream behind it (round 6: measured on the model rank of the
    # scaling model, whose process keeps two contexts alive: 2.4 ms of a 12.5 ms cycle).  Ask for 8 unless the user has set
    # the variable; it is read when the HIP runtime initialises, i.e. before the first HIP call of the process.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.femo_abi_version() != ABI_VERSION:
        raise FemoError("libfemo_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().femo_last_error()
        raise FemoError(msg.decode() if msg else f"libfemo_hip error {rc}")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().femo_device_count(C.byref(n))
    return n.value if rc == 0 else 0
