"""ctypes binding of ``libotvae_hip.so`` (C ABI in ``include/otvae.h``).

The product path has no CPU implementation: if the library is missing or a call fails, an exception is raised.
Bad shapes surface as ``ValueError`` like the reference's argument validation (e.g. ot/w2_utils.py:619-707).
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OTVAE_LIB (a file name next to this module) selects another build of the same ABI: A/B timing of kernel variants on one box
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("OTVAE_LIB", "libotvae_hip.so")))

_lock = threading.Lock()
_lib = None

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
pp = C.POINTER(C.c_void_p)  # host array of device pointers
pi32 = C.POINTER(C.c_int)
pu32 = C.POINTER(C.c_uint32)


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "Hs", "Ws", "Cs", "up", "Ho", "Wo", "Cn", "KH", "KW", "stride", "pad")]


pg = C.POINTER(ConvGeom)

JOB_FWD, JOB_BWD_DATA, JOB_BWD_WEIGHT = 0, 1, 2
DEFER_DENSE, DEFER_SPARSE = 1, 2  # otvae_conv_job.defer_reduce (include/otvae.h)


class BnFold(C.Structure):  # otvae_bn_fold
    _fields_ = [("slots", C.c_void_p), ("ld", C.c_int32), ("nslots", C.c_int32), ("count", C.c_int64), ("eps", C.c_float),
                ("momentum", C.c_float)] + [(n, C.c_void_p) for n in ("gamma", "beta", "running_mean", "running_var",
                                                                      "num_batches_tracked", "mean_out", "invstd_out", "scale_out",
                                                                      "shift_out")]


class ConvJob(C.Structure):  # otvae_conv_job
    _fields_ = ([(n, C.c_int32) for n in ("kind", "relu", "has_bias", "defer_reduce")] + [("geom", ConvGeom)] +
                [(n, C.c_void_p) for n in ("x", "scale", "shift", "w", "bias", "residual", "y", "stat_partial", "gy",
                                           "mean", "invstd", "gv", "bn_partial", "wpartial", "gw", "gb", "stat_slots",
                                           "bn_slots")] +
                [("stat_nslots", C.c_int32), ("bn_nslots", C.c_int32), ("fold", BnFold)])


pj = C.POINTER(ConvJob)

# name -> (restype, argtypes); mirrors include/otvae.h one to one (tests/test_abi.py checks both directions)
SIGNATURES = {
    "otvae_abi_version": (i32, []),
    "otvae_last_error": (C.c_char_p, []),
    "otvae_device_info": (i32, [pi32, pi32, C.c_char_p, i32]),
    "otvae_stream_create": (i32, [pp]),
    "otvae_stream_destroy": (i32, [vp]),
    "otvae_bn_stats_nparts": (i32, [i64, i32]),
    "otvae_bn_stats": (i32, [vp, i64, i32, vp, vp]),
    "otvae_bn_slots_words": (i64, [i32, i32]),
    "otvae_bn_stats_slots": (i32, [vp, i64, i32, vp, i32, i32, vp]),
    "otvae_bn_bwd_apply_slots": (i32, [i32, pp, vp, pp, pi32, i32, i64, i32, vp, vp, pp, pp, pp, i32, vp, vp]),
    "otvae_bn_finalize_slots": (i32, [i32, C.POINTER(BnFold), i32, vp]),
    "otvae_bn_finalize": (i32, [vp, i32, i32, i64, i32, f32, f32, vp, vp, i32, pp, pp, pp, pp, pp, pp, pp, vp]),
    "otvae_conv_fwd_stats_ws": (i32, [pg, pi32, pi32]),
    "otvae_conv_fwd": (i32, [pg, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "otvae_weight_transpose": (i32, [vp, vp, i32, i32, i32, vp]),
    "otvae_weight_transpose_batched": (i32, [vp, vp, vp, i32, i64, vp]),
    "otvae_conv_bwd_data_ws": (i32, [pg, pi32, pi32]),
    "otvae_conv_bwd_data": (i32, [pg, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "otvae_bn_bwd_finalize": (i32, [i32, pp, pi32, i32, i64, i32, vp, vp, pp, pp, pp, vp, vp]),
    "otvae_bn_bwd_apply": (i32, [i32, pp, vp, vp, i64, i32, vp, vp]),
    "otvae_conv_bwd_weight_ws": (i32, [pg, i32, pi32]),
    "otvae_conv_bwd_weight": (i32, [pg, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp]),
    "otvae_wgrad_reduce_batched": (i32, [i32, pp, pi32, pi32, pi32, pi32, pp, pp, pi32, pu32, vp]),
    "otvae_conv_dead_taps": (i32, [pg, pu32]),
    "otvae_gmm_diag_energy": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "otvae_conv_multi": (i32, [i32, pj, vp]),
    "otvae_conv_multi_last": (i32, [pu32, pi32]),
    "otvae_attn_fwd": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "otvae_attn_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "otvae_attn_fwd_scaled": (i32, [vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "otvae_attn_bwd_scaled": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]),
    "otvae_attn_stage_plan": (i32, [i32, i32, i32, i32, i32, vp]),
    "otvae_attn_stage_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "otvae_attn_stage_fwd_fold": (i32, [vp, C.POINTER(BnFold), vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "otvae_attn_stage_bwd_slots": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, i32, vp]),
    "otvae_attn_stage_bwd_plan": (i32, [i32, i32, i32, i32, vp]),
    "otvae_attn_stage_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "otvae_attn_dropout_fwd": (i32, [vp, i32, i32, i32, i32, f32, f32, i32, vp, i32, vp, vp, vp, vp]),
    "otvae_attn_dropout_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, vp]),
    "otvae_attn_dropout_mask": (i32, [i32, i32, i32, f32, vp, vp, vp]),
    "otvae_attn_cross_fwd": (i32, [vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, f32, vp, i32, vp, vp, vp, vp]),
    "otvae_attn_cross_mask": (i32, [i32, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_attn_cross_bwd": (i32, [vp, i64, i32, vp, vp, i64, i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp, vp, i64, i32,
                                   vp, vp, i64, i32, vp]),
    "otvae_layernorm_fwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp]),
    "otvae_layernorm_bwd_ws": (i32, [i32, i32]),
    "otvae_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "otvae_layernorm_dropout_fwd": (i32, [vp, vp, vp, vp, i32, i32, f32, f32, vp, i32, vp, vp, vp, vp, vp, vp]),
    "otvae_layernorm_dropout_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "otvae_layernorm_dropout_mask": (i32, [i32, i32, f32, vp, vp, vp]),
    "otvae_dropout_fwd": (i32, [vp, i64, i32, i32, f32, vp, i32, vp, vp, vp]),
    "otvae_dropout_bwd": (i32, [vp, vp, i64, i32, i32, f32, vp, vp, vp]),
    "otvae_gaussian_prior_fwd": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_gaussian_prior_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp, vp]),
    "otvae_gaussian_prior_ex_fwd": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp]),
    "otvae_gaussian_prior_ex_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]),
    "otvae_gaussian_prior_cond_fwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "otvae_gaussian_prior_cond_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "otvae_gaussian_prior_cond_ex_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp]),
    "otvae_gaussian_prior_cond_ex_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "otvae_copy_batched": (i32, [i32, vp, vp, vp, vp]),
    "otvae_nelbo_ws": (i32, []),
    "otvae_nelbo_fwd": (i32, [vp, vp, i64, vp, i32, f32, vp, vp, vp]),
    "otvae_nelbo_bwd": (i32, [vp, vp, i64, i32, f32, vp, vp, vp, vp]),
    "otvae_bn_act_fwd": (i32, [vp, vp, vp, i32, i64, i32, vp, vp]),
    "otvae_bn_act_bwd_parts": (i32, [i64]),
    "otvae_bn_act_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, vp, vp]),
    "otvae_scale_f32": (i32, [vp, f32, i64, vp, vp]),
    "otvae_weight_expand_fwd": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "otvae_weight_expand_bwd": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "otvae_group_norm_act_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "otvae_group_norm_act_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "otvae_colsum_f32": (i32, [vp, i32, i32, vp, vp]),
    "otvae_embedding_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "otvae_film_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "otvae_film_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "otvae_dropout2d_fwd": (i32, [vp, i32, i32, i32, f32, vp, i32, vp, vp, vp]),
    "otvae_dropout2d_bwd": (i32, [vp, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_dropout2d_mask": (i32, [i32, i32, f32, vp, vp, vp]),
    "otvae_mvn_logprob_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "otvae_mvn_logprob_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "otvae_step_begin": (i32, [vp, vp]),
    "otvae_adam_step": (i32, [vp, vp, vp, vp, i64, vp, vp, f32, vp]),
    "otvae_adam_step_dev": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "otvae_adam_step_ema": (i32, [vp, vp, vp, vp, i64, vp, vp, f32, vp, vp, vp, vp, vp, i64, vp, f64, vp]),
    "otvae_ema_update": (i32, [vp, vp, i64, f64, vp]),
    "otvae_adam_step_guarded": (i32, [vp, vp, vp, vp, i64, vp, vp, f32, vp, vp, vp, vp, vp, i64, vp]),
    "otvae_step_begin_guarded": (i32, [vp, vp, vp, i64, vp]),
    "otvae_step_begin_slots": (i32, [vp, vp, vp, i64, vp, i64, vp]),
    "otvae_zero_words": (i32, [vp, i64, vp]),
    "otvae_grad_clip_ws": (i32, []),
    "otvae_grad_clip_coef": (i32, [vp, i64, f32, f32, vp, vp, vp]),
    "otvae_conv_generic_fwd": (i32, [pg, vp, vp, vp, vp, vp]),
    "otvae_conv_generic_bwd_data": (i32, [pg, vp, vp, vp, vp]),
    "otvae_conv_generic_bwd_weight_ws": (i64, [pg, i32]),
    "otvae_conv_generic_bwd_weight": (i32, [pg, vp, vp, i32, vp, vp, vp, vp]),
    "otvae_sinkhorn_ws": (i64, [i32, i32, i32, i32]),
    "otvae_sinkhorn_log": (i32, [i32, vp, vp, vp, i32, i32, i32, f64, i32, f64, vp, vp, vp, vp, vp, vp]),
    "otvae_sinkhorn_tape_bytes": (i64, [i32, i32, i32, i32, i32]),
    "otvae_sinkhorn_bwd_ws": (i64, [i32, i32, i32, i32, i32]),
    "otvae_sinkhorn_log_tape": (i32, [i32, vp, vp, vp, i32, i32, i32, f64, i32, f64, vp, vp, vp, vp, vp, vp]),
    "otvae_sinkhorn_log_bwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, f64, i32, vp, vp, vp, vp, vp, vp]),
    "otvae_sinkhorn_log_normalized": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, f64, i32, f64, vp, vp, vp, vp, vp, vp, vp]),
    "otvae_sinkhorn_prior_ws": (i64, [i32, i32, i32]),
    "otvae_sinkhorn_prior_fwd": (i32, [i32, vp, vp, i32, i32, i32, f64, i32, f64, f64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "otvae_sqdist_max_parts": (i32, [i32, i32, i32]),
    "otvae_sqdist_max": (i32, [i32, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "otvae_normal_fill": (i32, [vp, i64, vp, i32, i32, vp]),
    "otvae_ot_cost": (i32, [i32, vp, vp, i32, i32, i32, vp, vp, vp]),
    "otvae_sqdist": (i32, [i32, vp, vp, i32, i32, i32, i32, vp, vp]),
    "otvae_ot_cost_grad": (i32, [i32, vp, vp, vp, vp, i32, f64, vp, i32, i32, i32, vp, vp]),
    "otvae_w2_prior_tail": (i32, [vp, vp, vp, vp, vp, vp, i32, f64, i32, vp, vp, vp]),
    "otvae_w2_prior_bwd": (i32, [i32, vp, i32, i32, vp, vp, vp, vp, i32, f64, vp, vp, vp]),
    "otvae_gauss_stats_ws": (i64, [i32, i32, i32, i32]),
    "otvae_gauss_stats": (i32, [i32, vp, i32, i32, i32, i32, i32, f64, vp, vp, vp, vp, vp]),
    "otvae_mean_cov": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "otvae_eigh_ws": (i64, [i32, i32]),
    "otvae_eigh_onesided_ws": (i64, [i32, i32]),
    "otvae_eigh_block_onesided_ws": (i64, [i32, i32]),
    "otvae_eigh_fn": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "otvae_eigh_fn_warm": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "otvae_make_psd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "otvae_cholesky": (i32, [vp, i32, i32, vp, vp, vp]),
    "otvae_w2_transport_ws": (i64, [i32, i32]),
    "otvae_w2_transport": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f64, i32, vp, vp, vp, vp, vp, vp]),
    "otvae_gemm_f64": (i32, [i32, i32, i32, i32, i32, i32, f64, vp, i32, vp, i32, f64, vp, vp]),
    "otvae_gemm_f32": (i32, [i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, i32, f32, vp, vp]),
    "otvae_softmax_rows": (i32, [i32, vp, i64, i32, f64, vp, vp]),
    "otvae_softmax_rows_bwd": (i32, [i32, vp, vp, i64, i32, f64, vp, vp]),
    "otvae_lse_rows": (i32, [i32, vp, i64, i32, vp, vp]),
    "otvae_lse_rows_bwd": (i32, [i32, vp, vp, vp, i64, i32, vp, vp]),
    "otvae_w2_tail": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "otvae_apply_transport": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "otvae_codebook_assign": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_codebook_probs": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_codebook_probs_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]),
    "otvae_codebook_probs_bwd_atoms": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "otvae_codebook_energy": (i32, [vp, vp, i32, i32, i32, i32, i32, f32, vp, vp]),
    "otvae_codebook_energy_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "otvae_codebook_kmeans": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
}


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Loads (once) and returns the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m ot_vae_lightning_amd.build` "
                "(there is no CPU fallback for the MI355X path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    return load().otvae_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = f"{what} failed (code {rc}): {last_error()}"
    if rc == -1:
        raise ValueError(msg)
    if rc == -2:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def _host_tensor(t):
    raise RuntimeError(f"a {tuple(t.shape)} {t.dtype} tensor on {t.device} was about to be handed to an MI355X kernel: move the "
                       "module / tensor to the GPU first (.cuda()); there is no CPU path")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  A host tensor is refused here, at the last gate before a kernel: its address
    would fault the GPU (and may take the node's other GPUs with it)."""
    if t is None:
        return None
    if not t.is_cuda:
        _host_tensor(t)
    return t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers for the `const float* const*` style arguments."""
    arr = (C.c_void_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        if t is not None:
            if not t.is_cuda:
                _host_tensor(t)
            arr[i] = t.data_ptr()
    return arr


def stream():
    """Raw handle of torch's current HIP stream on the current device (torch.cuda.current_stream() builds a Stream object per
    call: 10 us of host time, ~90 times per eagerly issued step)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


# ---- streams of our own ------------------------------------------------------------------------------------------------------------
# torch.cuda.Stream() does not create a stream: it takes the next of a pool of 32 per device, round-robin.  A process that has built a
# few engines (each with a capture stream, a warm-up stream, a reducer stream ...) gets, as its "new" side stream, the very queue its
# capture stream or its prior lane already is -- a captured step whose fork waits on itself; the HIP graph executor then died in
# hip::Graph::UpdateStreams at the first replay (found in round 4 when the test suite grew past that count).  Every side / lane /
# capture / warm-up stream of this package therefore is a HIP stream of the library's own (otvae_stream_create), wrapped as a
# torch.cuda.ExternalStream; a wrapper that dies hands its stream back to a free list (never destroyed: at most as many as were ever
# alive at once).
_free_streams: dict = {}   # device index -> [raw handles]


class _OwnStream(torch.cuda.ExternalStream):
    """ExternalStream over a stream of otvae_stream_create; hands the handle back when it dies.  (``__del__``, not ``weakref.finalize``:
    a weak reference to a torch stream object crashed the interpreter's finalization, _PyWeakref_ClearRef, on torch 2.10.)"""

    def __del__(self):
        try:
            _free_streams.setdefault(self._otvae_index, []).append(self._otvae_handle)
        except Exception:   # interpreter shutdown: the module's globals are gone, and so is the need
            pass


def fresh_stream(device=None) -> "torch.cuda.Stream":
    """A stream no other live stream of this process aliases (see above)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    free = _free_streams.setdefault(index, [])
    if free:
        handle = free.pop()
    else:
        out = C.c_void_p()
        with torch.cuda.device(index):
            check(load().otvae_stream_create(C.byref(out)), "otvae_stream_create")
        handle = out.value
    s = _OwnStream(handle, device=torch.device("cuda", index))
    s._otvae_index, s._otvae_handle = index, handle
    return s


def require_cuda(t: torch.Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the MI355X (got device {t.device}); "
                           "ot_vae_lightning_amd has no CPU execution path")
