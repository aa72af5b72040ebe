"""ctypes binding of libvitk.so (declared in include/vitk.h).

The library is the only compute path of this package: if it is missing, loading fails loudly with
instructions to build it; there is no PyTorch/CPU fallback for any op.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# torch must be imported before libvitk.so is dlopen'ed: both need libamdhip64, and the process has to end up with ONE HIP
# runtime (the one torch ships and initialises); loading ours first pulls in /opt/rocm's copy and the device is not found.
import torch  # noqa: F401

import os

# VITK_LIB_PATH points at another build of the same library (A/B timing of a kernel change); default: the in-tree build
LIB_PATH = Path(os.environ.get("VITK_LIB_PATH") or (Path(__file__).resolve().parent / "csrc" / "libvitk.so"))

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_i64 = C.c_int64
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/vitk.h one to one
SIGNATURES = {
    "vitk_version": (_i, []),
    "vitk_last_error": (C.c_char_p, []),
    "vitk_prof_enable": (_i, [_i]),
    "vitk_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "vitk_prof_reset": (_i, []),
    "vitk_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "vitk_layernorm_fwd_gather": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _i, _p]),
    "vitk_layernorm_bwd_gather": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vitk_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "vitk_layernorm_bwd_reduce": (_i, [_p, _sz, _i, _i, _p, _p, _i, _p]),
    "vitk_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vitk_layernorm_bwd_noalloc": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vitk_layernorm_bwd_scaled": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vitk_gemm_nt": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _p]),
    "vitk_gemm_nt_res_scatter": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p]),
    "vitk_gemm_nt_sched_reset": (_i, [_p]),
    "vitk_gemm_nt_sched_poke": (_i, [_p, _i, C.c_uint]),
    "vitk_gemm_nt_sched_state": (_i, [_p, C.POINTER(C.c_uint), C.POINTER(_i)]),
    "vitk_gemm_tn_workspace": (_sz, [_i, _i, _i]),
    "vitk_gemm_tn": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    "vitk_attention_fwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "vitk_attention_bwd_workspace": (_sz, [_i, _i, _i]),
    "vitk_attention_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _sz, _p]),
    "vitk_attention_generic_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "vitk_attention_generic_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "vitk_im2col_patches": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vitk_prefix_tokens_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vitk_pos_tokens_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vitk_pool_avg_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vitk_pool_avg_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vitk_gather_rows": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vitk_scatter_rows": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vitk_slice_tokens": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vitk_copy_samples": (_i, [_p, _p, _p, _i, _i64, _p]),
    "vitk_rowscale_gather": (_i, [_p, _p, _i, _p, _i, _p, _i, _p]),
    "vitk_rowscale": (_i, [_p, _p, _i, _p, _i, _i, _p]),
    "vitk_cast_f32_to_bf16": (_i, [_p, _p, _i64, _p]),
    "vitk_cast_bf16_to_f32": (_i, [_p, _p, _i64, _p]),
    "vitk_transpose_bf16": (_i, [_p, _p, _i, _i, _p]),
    "vitk_transpose_bf16_batched": (_i, [_p, _i, _i, _p]),
    "vitk_add_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "vitk_copy_from_host": (_i, [_p, _p, _i64, _p]),
    "vitk_preprocess_images": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p, _p]),
    "vitk_preprocess_patches": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p, _p]),
    "vitk_resample_images": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _f, _f, _p, _p, _p]),
    "vitk_resample_patches": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p, _p]),
    "vitk_mixup_batch": (_i, [_p, _i, _i64, _f, _f, _p]),
    "vitk_cutmix_batch": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vitk_mixup_targets": (_i, [_p, _p, _i, _i, _i, _f, _f, _f, _f, _p]),
    "vitk_mixup_elem": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "vitk_mixup_targets_elem": (_i, [_p, _p, _i, _i, _i, _p, _f, _f, _p]),
    "vitk_xent_label_fwd_bwd": (_i, [_p, _i, _p, _f, _p, _p, _i, _f, _i, _i, _p]),
    "vitk_xent_soft_fwd_bwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _f, _i, _i, _p]),
    "vitk_kd_kl_fwd_bwd": (_i, [_p, _i, _p, _i, _f, _p, _p, _i, _f, _i, _i, _p]),
    "vitk_argmax_rows": (_i, [_p, _i, _p, _i, _i, _p]),
    "vitk_adamw_step": (_i, [_p, _p, _p, _p, _p, _p, _d, _i64, _d, _d, _d, _d, _d, _i, _f, _p]),
    "vitk_ema_update": (_i, [_p, _p, _d, _i64, _p]),
    "vitk_clip_grad_scale": (_i, [_p, _i64, _p, _f, _p]),
    "vitk_sumsq": (_i, [_p, _i64, _p, _i, _p, _sz, _p]),
    "vitk_sumsq_workspace": (_sz, [_i64]),
}

EPI_PLAIN, EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_PATCH, EPI_DGELU, EPI_BIAS_GELU_DG, EPI_MUL = range(9)


class VitkError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libvitk.so and attach the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VitkError(
            f"{LIB_PATH} not found: build the HIP kernels first "
            "(python -m vision_transformers_torch_xla_amd.build, or __graft_entry__.build()). "
            "There is no fallback implementation.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vitk_last_error()
        raise VitkError(f"{what or 'vitk call'} failed (rc={rc}): {msg.decode() if msg else ''}")
