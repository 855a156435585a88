"""ctypes binding of ``libbasd_hip.so`` (C ABI declared in ``include/basd_hip.h``).

There is deliberately no CPU fallback: if the library is missing, or a call
returns a non-zero status, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbasd_hip.so")

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> argtypes (return type is always int)
SIGNATURES = {
    "basd_gemm_nt": [vp, i32, i64, i64, i64, i32, i64, vp, i64, i64, i32, i32, i32, i32, vp, i64, i64, f32, vp, f32,
                     vp, vp, vp],
    "basd_gemm_tn_splits": [i32],
    "basd_gemm_tn": [vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp,
                     vp, i64, i64, f32, vp],
    "basd_colmean_parts": [i32],
    "basd_colmean": [vp, i32, i64, i64, i64, i32, i64, i32, i32, i32, i32, vp, vp, vp],
    "basd_colmean_multi": [vp, i32, i64, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp],
    "basd_syrk_splits": [i32, i32, i32],
    "basd_syrk_multi": [vp, i32, i64, i64, i64, i32, i32, i32, i32, vp, vp, i32, vp, vp, i64, i32, vp, i32, i32, vp],
    "basd_jacobi_workspace_ints": [i32, i32],
    "basd_jacobi_twopass_workspace_bytes": [i32, i32, i32],
    "basd_jacobi_stacked_twopass": [vp, i64, i32, i32, vp, i32, i32, f32, vp, vp, vp],
    "basd_jacobi_tuning": [i32],
    "basd_jacobi_ordering": [i32],
    "basd_gemm_tuning": [i32],
    "basd_gemm_tuning_get": [],
    "basd_jacobi_onesided": [vp, i64, i32, i32, i32, i32, vp, vp, i32, i32, f32, vp, vp, vp],
    "basd_sort_extract": [vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp],
    "basd_tridiag_workspace_bytes": [i32, i32],
    "basd_tridiag_tuning": [i32, i32, i32, i32, i32, i32],
    "basd_tridiag": [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp],
    "basd_tridiag_ranked": [vp, i64, i32, i32, vp, vp, vp, vp, vp, i32, f64, i32, vp, vp, vp, vp],
    "basd_tridiag_ranked_gated": [vp, i64, i32, i32, vp, vp, vp, vp, vp, i32, f64, i32, vp, vp, vp, C.c_uint, i32, vp],
    "basd_flag_set": [vp, C.c_uint, vp],
    "basd_scale_unless_one": [vp, i64, vp, vp, vp],
    "basd_event_create": [vp],
    "basd_event_destroy": [vp],
    "basd_stream_wait_event": [vp, vp],
    "basd_tridiag_eigenvalues": [vp, vp, i32, i32, vp, vp],
    "basd_tridiag_apply_q": [vp, vp, i32, i32, i32, vp, vp, i32, i32, vp],
    "basd_tridiag_shifted_solve": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "basd_tridiag_mp_rank": [vp, vp, i32, i32, f64, i32, vp, vp, vp, vp, vp],
    "basd_tridiag_mp_rank_rank1": [vp, vp, vp, i32, i32, f64, f64, i32, vp, vp, vp, vp],
    "basd_tridiag_eigenvectors": [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp],
    "basd_mp_rank": [vp, i32, i32, f64, i32, vp, vp, vp],
    "basd_grassmann_distance": [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp],
    "basd_grassmann_distance_padded": [vp, i32, i32, vp, vp, i32, vp, i32, vp, vp],
    "basd_selector_tail": [vp] * 10 + [i32, i32, i32, i32] + [vp] * 14 + [vp],
    "basd_sqrt_clamp": [vp, vp, i64, vp],
    "basd_gram_finish": [vp, vp, i32, i32, i64, vp, vp, vp],
    "basd_cross_entropy": [vp, i32, i64, i32, i32, vp, vp, i64, f32, i64, vp, vp, vp],
    "basd_token_weights": [vp, i32, vp, i32, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp,
                           vp, vp, vp, vp, vp, vp],
    "basd_student_project": [vp, i32, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "basd_teacher_center": [vp, i32, vp, i32, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "basd_teacher_center_multi": [vp, i32, vp, i32, i32, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "basd_teacher_center_stream_scratch_floats": [i32, i32, i32, i32],
    "basd_teacher_center_stream": [vp, i32, vp, i32, i32, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "basd_gram_f64": [vp, i64, i32, i32, i32, vp, i64, vp],
    "basd_gram_f64_split": [vp, i64, i32, i32, i32, i32, vp, vp, vp],
    "basd_chol_f64": [vp, i64, i32, i32, vp, i64, vp],
    "basd_stack_product": [vp, vp, i64, i32, i32, i32, vp, i64, vp],
    "basd_procrustes_finalize": [vp, i64, vp, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                                 vp, vp],
    "basd_student_project_multi": [vp, i32, i64, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp,
                                   vp, vp, vp],
    "basd_student_grad_multi": [vp, i32, i64, i64, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, f32, vp,
                                vp, vp, vp],
    "basd_student_grad_fused": [vp, i32, i64, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp,
                                f32, vp, vp],
    "basd_procrustes_forward_fused": [vp, vp],
    "basd_procrustes_tuning": [i32],
    "basd_jacobi_plain4_fits": [i32],
    "basd_stack_product_t": [vp, vp, i64, i32, i32, i32, vp, i64, vp],
    "basd_kprime_from_transposed": [vp, i64, vp, i32, i32, vp, i64, i32, vp, i64, vp, vp],
    "basd_ustack_stash": [vp, i64, i32, i32, vp, i64, vp],
    "basd_ustack_from_transposed": [vp, i64, vp, i32, i32, vp, i64, vp, i64, vp, i32, vp, vp],
    "basd_resample_tokens": [vp, i32, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "basd_resample_tokens_adjoint": [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "basd_student_grad": [vp, i32, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp],
    "basd_teacher_factor": [vp, i64, vp, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "basd_teacher_factor_tiled": [vp, i64, vp, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "basd_mix_grad_tokens": [vp, vp, i32, i32, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "basd_mix_grad_tokens_scratch_floats": [i32, i32, i32, i32],
    "basd_mix_grad_tokens_onepass": [vp, vp, i32, i32, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "basd_token_weight_bwd": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, i64, i32,
                              i32, i32, vp, vp, vp],
    "basd_build_angle_stack": [vp, i32, vp, i32, vp, vp],
    "basd_grassmann_distance_bwd": [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp],
    "basd_eigvec_k2": [vp, vp, i32, i32, i32, vp, vp],
    "basd_selector_chain": [vp],
    "basd_rank_certificate": [vp, vp, i32, i32, f64, vp, vp, vp],
    "basd_rank_certificate_scratch_bytes": [i32],
    "basd_debug_fill_lds": [C.c_uint, vp],
    "basd_selector_chain_tail": [vp, i32, i32],
    "basd_jacobi_lds_square_fits": [i32],
    "basd_event_record": [vp, vp],
    "basd_event_synchronize": [vp],
    "basd_event_query": [vp],
    "basd_stream_create_priority": [vp, i32],
    "basd_event_create_timed": [vp],
    "basd_event_elapsed_ms": [vp, vp, vp],
}

class ProcrustesArgs(C.Structure):
    """BasdProcrustesArgs of include/basd_hip.h (every field 8 bytes wide)."""
    _PTRS = ("student_ptrs", "student_host_ptrs")
    _fields_ = (
        [("student_ptrs", vp), ("student_host_ptrs", vp)]
        + [(n, i64) for n in ("s_dtype", "s_sb", "s_sn", "s_aligned")]
        + [("tok_ptrs", vp)] + [(n, i64) for n in ("t_dtype", "t_sb", "t_sn", "t_sd")]
        + [("attn_ptrs", vp)] + [(n, i64) for n in ("a_dtype", "a_sb", "a_sh", "a_sq", "a_sk")]
        + [("mix", vp)]
        + [(n, i64) for n in ("E", "L", "G", "B", "n_s", "n_t", "d_s", "d_t", "H", "A", "has_cls", "n_a", "n",
                              "max_sweeps")]
        + [(n, vp) for n in ("atap0", "atap1", "alam", "tap0", "tap1", "lam", "range0", "range1", "g0", "g1", "glam",
                             "omega", "omega_t", "raw", "mu_t", "tc", "mu_s", "tr_part", "tr_s", "a_prime", "g_all",
                             "l_all", "W", "sigma", "jflags", "sweeps", "tr_t", "nuc", "loss_b", "k_prime", "h", "dx",
                             "grad_layers", "g_slabs")]
        + [("g_splits", i64)]
        + [("uw_ce", vp), ("uw_out", vp), ("jac_ws", vp), ("w_stack", vp), ("sigma_u", vp)]
    )


class SelectorChainArgs(C.Structure):
    """BasdSelectorChain of include/basd_hip.h (every field 8 bytes wide)."""
    _fields_ = (
        [("teacher_host_ptrs", vp)] + [(n, i64) for n in ("t_dtype", "t_sb", "t_sn", "t_sd")]
        + [("student_ptrs", vp)] + [(n, i64) for n in ("s_dtype", "s_sb", "s_sn", "s_sd", "s_vec_ok")]
        + [("proj_t", vp), ("proj_s_t", vp)]
        + [(n, i64) for n in ("E", "L", "B", "n_s", "n_t", "d_s", "d_t")]
        + [("mp_factor", f64)]
        + [(n, i64) for n in ("rank_cap", "kmax", "kmax_cap", "mode")]
        + [(n, vp) for n in ("z", "z_sums", "z_ptrs", "z_means", "t_slabs")] + [("t_splits", i64)]
        + [(n, vp) for n in ("s_partial", "s_means", "s_slabs")] + [("s_splits", i64), ("s_parts", i64)]
        + [(n, vp) for n in ("grams", "d", "e", "tau", "vh", "vals", "tri_work", "tri_work_s", "ranks", "host_mirror",
                             "student_status_mirror", "zv", "vecs", "u_rot", "sw", "cos", "sigma", "d_out", "k_arr",
                             "sw_index", "jflags", "main_stream", "chain_stream", "student_stream", "tail_stream",
                             "ev_fork", "ev_student", "ev_ranks", "ev_tail", "ev_slot_free", "ev_tgram", "ev_tg0",
                             "release_delay", "fact_stream", "go_flag", "go_value", "go_budget",
                             "tm_proj", "tm_tgram", "tm_scol0", "tm_scol1", "tm_sgram", "tm_tri0", "tm_mid", "tm_spec",
                             "cert_stream", "cert_mirror", "ev_cert", "cert_scratch")]
    )


EINVAL, EUNSUPPORTED = -1, -2        # BASD_EINVAL / BASD_EUNSUPPORTED of include/basd_hip.h

# sizing helpers declared `long` in include/basd_hip.h
LONG_RESULTS = {"basd_tridiag_workspace_bytes", "basd_jacobi_twopass_workspace_bytes",
                "basd_mix_grad_tokens_scratch_floats", "basd_teacher_center_stream_scratch_floats",
                "basd_rank_certificate_scratch_bytes"}

_lock = threading.Lock()
_lib = None


def load() -> C.CDLL:
    """Load the library once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} not found: build it with `make -C vit-inductive-bias-distillation_amd/csrc` "
                    "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = C.c_long if name in LONG_RESULTS else C.c_int
            _lib = lib
    return _lib


# Optional per-entry-point timing (bench.py): name -> list of (start_event, end_event) recorded on the
# stream the kernels are queued on.  `timed_names` None = every entry point.
timing: dict | None = None
timed_names: set | None = None


def call(name: str, *args) -> None:
    """Call an entry point that reports a status; non-zero becomes RuntimeError."""
    if timing is not None and (timed_names is None or name in timed_names):
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        status = getattr(load(), name)(*args)
        e1.record()
        timing.setdefault(name, []).append((e0, e1))
    else:
        status = getattr(load(), name)(*args)
    if status != 0:
        kind = "invalid argument / unsupported shape" if status < 0 else "HIP error"
        raise RuntimeError(f"{name} failed with status {status} ({kind})")


def query(name: str, *args) -> int:
    """Call a pure host-side sizing helper (returns a count, not a status)."""
    return int(getattr(load(), name)(*args))
