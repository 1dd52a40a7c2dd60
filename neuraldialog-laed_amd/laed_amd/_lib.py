"""ctypes binding of liblaed_hip.so (the C ABI declared in include/laed_hip.h).

There is NO CPU fallback: if the shared library is missing, import fails loudly with the build command.
All wrappers take torch CUDA tensors, check dtype/contiguity on the host, and enqueue on torch's current stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("LAED_HIP_LIB") or os.path.join(_HERE, "liblaed_hip.so")    # override: A/B builds of the library

if not os.path.exists(_SO):
    raise ImportError(
        "laed_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C neuraldialog-laed_amd/csrc` (hipcc, --offload-arch=gfx950). "
        "There is no CPU fallback for this package." % _SO)

lib = ctypes.CDLL(_SO)

_i, _i64, _f, _d, _p, _u64 = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p,
                              ctypes.c_uint64)

_SIGS = {
    "laed_version": ([], _i),
    "laed_last_error": ([], ctypes.c_char_p),
    "laed_sgemm": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p], _i),
    "laed_embed_gather": ([_p, _i64, _p, _i, _i, _p, _f, _p, _p], _i),
    "laed_embed_gather_bt": ([_p, _i64, _i, _i, _p, _i, _i, _p, _i, _f, _u64, _u64, _p, _p, _p, _p], _i),
    "laed_embed_scatter_add": ([_p, _i64, _p, _i, _i, _p, _f, _i64, _p, _p, _i64, _p], _i),
    "laed_embed_scatter_workspace": ([_i64, _i, _i], _i64),
    "laed_dropout_keep_mask": ([_p, _i64, _f, _u64, _u64, _p], _i),
    "laed_uniform": ([_p, _i64, _u64, _u64, _p], _i),
    "laed_gru_fwd": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i64, _p], _i),
    "laed_gru_fwd_cus": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i64, _i, _p], _i),
    "laed_gru_bwd_cus": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i64, _i, _p], _i),
    "laed_gru_persist_timeouts": ([ctypes.POINTER(ctypes.c_uint)], _i),
    "laed_gru_persist_launches": ([ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)], _i),
    "laed_gru_fwd_workspace": ([_i, _i, _i, _i], _i64),
    "laed_gru_bwd": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i64, _p], _i),
    "laed_gru_bwd_workspace": ([_i, _i, _i, _i], _i64),
    "laed_sgemm_ws": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i64, _p], _i),
    "laed_gumbel_latent_fwd": ([_p, _p, _i64, _i, _f, _i, _p, _p, _p, _p], _i),
    "laed_gumbel_latent_fwd_rng": ([_p, _i64, _i, _f, _i, _u64, _u64, _p, _p, _p, _p], _i),
    "laed_gumbel_latent_bwd": ([_p, _p, _p, _p, _i64, _i, _f, _p, _p], _i),
    "laed_latent_stats_blocks": ([], _i),
    "laed_latent_stats": ([_p, _i64, _i, _i, _p, _p, _p], _i),
    "laed_latent_stats_noalloc": ([_p, _i64, _i, _i, _p, _p], _i),
    "laed_latent_finalize": ([_p, _d, _i, _i, _p, _p, _p], _i),
    "laed_latent_stats_finalize": ([_p, _i64, _i, _i, _p, _p, _p, _p, _p], _i),
    "laed_latent_mi_coef": ([_p, _d, _i, _i, _p, _p], _i),
    "laed_latent_bwd": ([_p, _p, _p, _i, _i64, _d, _i, _i, _p, _i, _p], _i),
    "laed_avg_log_fwd": ([_p, _d, _i, _p, _p], _i),
    "laed_avg_log_coef": ([_p, _p, _d, _i, _p, _p], _i),
    "laed_cat_rows_blocks": ([], _i),
    "laed_cat_rows_fwd": ([_p, _p, _i64, _i, _i64, _i, _f, _p, _p, _p, _p], _i),
    "laed_cat_rows_bwd": ([_p, _p, _i64, _i, _i64, _i, _p, _f, _p, _p, _p], _i),
    "laed_proj_nll_ldl": ([_i], _i),
    "laed_proj_nll_fwd_workspace": ([_i64, _i], _i64),
    "laed_proj_nll_workspace": ([_i, _i, _i], _i64),
    "laed_proj_nll": ([_p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i, _i, _p, _i64, _i, _p, _p, _p, _p, _p, _p], _i),
    "laed_softmax_embed_workspace": ([_i, _i, _i], _i64),
    "laed_softmax_embed_fwd": ([_p, _p, _p, _p, _i64, _i, _i, _i, _p, _i64, _i, _p, _p, _p], _i),
    "laed_softmax_embed_bwd": ([_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _i64, _i, _p, _p, _p, _p, _i, _p], _i),
    "laed_log_softmax_rows": ([_p, _i64, _i, _i, _p, _p], _i),
    "laed_log_softmax_rows_bwd": ([_p, _i64, _p, _i64, _i64, _i, _p, _i64, _p], _i),
    "laed_argmax_rows": ([_p, _i64, _i, _i, _p, _p], _i),
    "laed_nll_reduce": ([_p, _p, _i64, _i64, _p, _p], _i),
    "laed_nll_prepare": ([_p, _i64, _i, _i, _i64, _i, _d, _p, _i, _p, _p, _p, _p], _i),
    "laed_nll_gather_fwd": ([_p, _i64, _p, _i64, _i, _i64, _p, _p], _i),
    "laed_nll_gather_bwd": ([_p, _p, _p, _i64, _i, _i64, _p, _i64, _p], _i),
    "laed_attn_score_fwd": ([_p, _p, _p, _i64, _i, _p, _p], _i),
    "laed_attn_score_bwd": ([_p, _p, _p, _i64, _i, _p, _p], _i),
    "laed_attn_pool_fwd": ([_p, _i64, _i64, _p, _i64, _i64, _i64, _i, _i, _p, _p, _p], _i),
    "laed_attn_pool_bwd": ([_p, _i64, _i64, _p, _p, _i64, _i, _i, _p, _i64, _i64, _p, _i64, _i64, _p], _i),
    "laed_tanh_fwd": ([_p, _p, _i64, _p], _i),
    "laed_tanh_bwd": ([_p, _p, _p, _i64, _p], _i),
    "laed_adam_step": ([_p, _p, _p, _p, _i64, _i, _d, _d, _d, _d, _f, _p], _i),
    "laed_adam_step_multi": ([_i, _p, _p, _p, _p, _p, _i, _d, _d, _d, _d, _f, _p], _i),
    "laed_sgd_step": ([_p, _p, _p, _i64, _d, _d, _f, _p], _i),
    "laed_rmsprop_step": ([_p, _p, _p, _p, _i64, _d, _d, _d, _d, _f, _p], _i),
    "laed_axpby": ([_p, _p, _i64, _f, _f, _p], _i),
    "laed_scale_by_dev": ([_p, _p, _p, _i64, _p], _i),
    "laed_scale_by_dev_multi": ([_i, _p, _p, _p, _p], _i),
    "laed_gate_functions": ([_p, _i64, _p, _p, _p], _i),
    "laed_prof_enable": ([_i], _i),
    "laed_prof_reserve": ([_i], _i),
    "laed_prof_collect": ([ctypes.POINTER(ctypes.c_double)], _i),
    "laed_set_gemm_mode": ([_i], _i),
    "laed_get_gemm_mode": ([], _i),
    "laed_set_xgemm_min_tiles": ([_i], _i),
    "laed_dot": ([_p, _p, _i64, _p, _p], _i),
    "laed_gru_cell_fwd": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p], _i),
    "laed_proj_argmax": ([_p, _p, _p, _i, _i, _i, _p, _p, _p], _i),
    "laed_gru_decode_greedy": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _p, _p, _p, _p, _p, _p], _i),
    "laed_gru_decode_sample": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _p, _u64, _u64,
                                _p, _p, _p, _p, _p, _p], _i),
}

EXPORTS = tuple(_SIGS.keys())

for _name, (_args, _res) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = _res


class LaedError(RuntimeError):
    pass


def check(status, what):
    if status != 0:
        msg = lib.laed_last_error()
        raise LaedError("%s failed with status %d: %s" % (what, status, msg.decode() if msg else ""))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LaedError("laed_amd kernels need CUDA (ROCm) tensors; got a CPU tensor")
    if t.dtype != dtype:
        raise LaedError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise LaedError("expected a contiguous tensor, got strides %s for shape %s" % (t.stride(), tuple(t.shape)))
    return ctypes.c_void_p(t.data_ptr())
