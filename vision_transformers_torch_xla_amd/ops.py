"""Tensor-level wrappers over the libvitk C ABI.

PyTorch is used only as the owner of device memory and of the HIP stream: every function here checks
device/dtype/contiguity, then hands raw device pointers, shapes and ``torch.cuda.current_stream()`` to one
``vitk_*`` entry point.  Nothing falls back to a torch op; CPU tensors are rejected.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_PATCH, EPI_PLAIN, check)

bf16 = torch.bfloat16
f32 = torch.float32


# torch.cuda.current_stream() builds a Stream object behind several Python-level device look-ups (8 us a call, a seventh of the
# host time of a launch-bound step); the raw handle of the current stream of the current device is one C call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise _lib.VitkError(f"{name}: expected a GPU tensor (the HIP kernels are the only implementation), got {t.device}")
    if t.dtype != dtype:
        raise _lib.VitkError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.VitkError(f"{name}: tensor must be contiguous")


class Workspace:
    """A growable device scratch buffer handed to kernels that need one (never allocated inside the library)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        nbytes = max(int(nbytes), 16)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


_default_ws = Workspace()


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps=1e-6, smap=None, rows_per_sample=0):
    """x bf16 [M,D] -> (y bf16 [M,D], mean fp32 [M], rstd fp32 [M]).  With smap (int32 [nk], sample indices): the outputs are
    compact over those samples, rows_per_sample rows each."""
    _chk(x, bf16, "x"); _chk(gamma, f32, "gamma"); _chk(beta, f32, "beta")
    M, D = x.shape
    lib = _lib.load()
    if smap is not None:
        _chk(smap, torch.int32, "smap")
        M = smap.numel() * rows_per_sample
    y = torch.empty((M, D), dtype=bf16, device=x.device)
    mean = torch.empty(M, dtype=f32, device=x.device)
    rstd = torch.empty(M, dtype=f32, device=x.device)
    if smap is not None:
        check(lib.vitk_layernorm_fwd_gather(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), M, D, eps, _ptr(smap),
                                            int(rows_per_sample), _stream()), "layernorm_fwd_gather")
    else:
        check(lib.vitk_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), M, D, eps, _stream()),
              "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dres=None, accumulate=False, ws: Workspace = None, dx=None,
                  rowscale=None, rows_per_group=0, in_inv=None, out_inv=None, out_rows=None, partial_out=None):
    """Returns dx bf16 [M,D] (= LN backward of dy, plus dres when given); writes/accumulates dgamma, dbeta (fp32 [D]).
    With rowscale (fp32 [M / rows_per_group]) returns (dx, dx_scaled) where dx_scaled[row] = rowscale[group] * dx[row].
    in_inv (int32 [samples]): dy / mean / rstd are compact, in_inv[sample] = compact index or -1 (sample skipped the branch);
    out_inv: dx_scaled is compact over the samples with out_inv >= 0 (out_rows rows).
    partial_out (a list): dgamma / dbeta are NOT written; their per-block partial sums go to a buffer of this call's own, which
    is appended to the list for layernorm_bwd_reduce (the caller runs it later, e.g. on another stream)."""
    _chk(dy, bf16, "dy"); _chk(x, bf16, "x"); _chk(gamma, f32, "gamma"); _chk(dgamma, f32, "dgamma"); _chk(dbeta, f32, "dbeta")
    if dres is not None:
        _chk(dres, bf16, "dres")
    M, D = x.shape
    lib = _lib.load()
    need = lib.vitk_layernorm_bwd_workspace(M, D)
    if partial_out is not None:
        w = torch.empty(max(int(need), 16), dtype=torch.uint8, device=x.device)
        partial_out.append((w, M, D, dgamma, dbeta, bool(accumulate)))
        dgamma = dbeta = None
    else:
        w = (ws or _default_ws).get(need, x.device)
    if dx is None:
        dx = torch.empty_like(x)
    dxs = None
    if rowscale is not None:
        _chk(rowscale, f32, "rowscale")
        dxs = torch.empty((M if out_inv is None else out_rows, D), dtype=bf16, device=x.device)
    if in_inv is not None or out_inv is not None:
        for t, n in ((in_inv, "in_inv"), (out_inv, "out_inv")):
            if t is not None:
                _chk(t, torch.int32, n)
        check(lib.vitk_layernorm_bwd_gather(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres), _ptr(dx), _ptr(rowscale),
                                            int(rows_per_group), _ptr(dxs), _ptr(in_inv), _ptr(out_inv), _ptr(dgamma), _ptr(dbeta),
                                            int(accumulate), M, D, _ptr(w), w.numel(), _stream()), "layernorm_bwd_gather")
        return (dx, dxs) if rowscale is not None else dx
    if rowscale is not None:
        check(lib.vitk_layernorm_bwd_scaled(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres), _ptr(dx), _ptr(rowscale),
                                            int(rows_per_group), _ptr(dxs), _ptr(dgamma), _ptr(dbeta), int(accumulate), M, D, _ptr(w),
                                            w.numel(), _stream()), "layernorm_bwd_scaled")
        return dx, dxs
    check(lib.vitk_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres), _ptr(dx), _ptr(dgamma),
                                 _ptr(dbeta), int(accumulate), M, D, _ptr(w), w.numel(), _stream()), "layernorm_bwd")
    return dx


def layernorm_bwd_reduce(item):
    """second stage of a layernorm_bwd call made with partial_out: item = (partial buffer, M, D, dgamma, dbeta, accumulate)"""
    w, M, D, dgamma, dbeta, accumulate = item
    check(_lib.load().vitk_layernorm_bwd_reduce(_ptr(w), w.numel(), M, D, _ptr(dgamma), _ptr(dbeta), int(accumulate), _stream()),
          "layernorm_bwd_reduce")


# ----------------------------------------------------------------------------------------------------- GEMM
def gemm_nt(a, b, *, epilogue=EPI_PLAIN, bias=None, out=None, out_dtype=bf16, res=None, rowscale=None, rows_per_group=0,
            out2=None, aux=None, pos=None, patches=0, prefix=0, out_rows=None, smap=None, rows_per_sample=0):
    """C[M,N] = a[M,K] @ b[N,K]^T with a fused epilogue (see vitk.h).  a, b bf16; 2-D, last dim contiguous."""
    for t, n in ((a, "a"), (b, "b")):
        if not t.is_cuda or t.dtype != bf16 or t.dim() != 2 or t.stride(1) != 1:
            raise _lib.VitkError(f"gemm_nt: {n} must be a 2-D bf16 GPU tensor with contiguous rows")
    M, K = a.shape
    N, Kb = b.shape
    if K != Kb:
        raise _lib.VitkError(f"gemm_nt: K mismatch {K} vs {Kb}")
    if out is None:
        rows = M if out_rows is None else out_rows
        out = torch.empty((rows, N), dtype=(f32 if epilogue == EPI_BIAS_F32 else out_dtype), device=a.device)
    if out.stride(-1) != 1:
        raise _lib.VitkError("gemm_nt: out rows must be contiguous")
    ldc = out.stride(0) if out.dim() == 2 else N
    ldres = res.stride(0) if res is not None else 0
    ldout2 = out2.stride(0) if out2 is not None else 0
    ldaux = aux.stride(0) if aux is not None else 0
    if smap is not None:        # bias + residual with the rows scattered by sample (stochastic depth over the kept samples only)
        if epilogue != EPI_BIAS_RES:
            raise _lib.VitkError("gemm_nt: a sample map goes with the bias + residual epilogue only")
        _chk(smap, torch.int32, "smap")
        check(_lib.load().vitk_gemm_nt_res_scatter(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), ldc, M, N, K, _ptr(bias), _ptr(res),
                                                   ldres, _ptr(rowscale), rows_per_group, _ptr(smap), int(rows_per_sample), _stream()),
              "gemm_nt_res_scatter")
        return out
    check(_lib.load().vitk_gemm_nt(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), ldc, M, N, K, epilogue, _ptr(bias),
                                   _ptr(res), ldres, _ptr(rowscale), rows_per_group, _ptr(out2), ldout2, _ptr(aux), ldaux,
                                   _ptr(pos), patches, prefix, _stream()), "gemm_nt")
    return out


def gemm_nt_sched_reset():
    """zero the persistent GEMM's tile counters of the current stream (in stream order) and clear a reported violation"""
    check(_lib.load().vitk_gemm_nt_sched_reset(_stream()), "gemm_nt_sched_reset")


def gemm_nt_sched_state():
    """(the 16 counter words of the current stream after synchronising with it, reporting mode, violation recorded)"""
    import ctypes
    words = (ctypes.c_uint * 16)()
    rep = ctypes.c_int(0)
    rc = _lib.load().vitk_gemm_nt_sched_state(_stream(), words, ctypes.byref(rep))
    if rc < 0:
        check(rc, "gemm_nt_sched_state")
    return list(words), rep.value, bool(rc)


def gemm_nt_sched_poke(word: int, value: int):
    """tests: write one of the 16 counter words of the current stream"""
    check(_lib.load().vitk_gemm_nt_sched_poke(_stream(), int(word), int(value)), "gemm_nt_sched_poke")


def gemm_tn(a, b, out, *, dbias=None, accumulate=False, ws: Workspace = None):
    """out[N,K] (fp32) (+)= a[M,N]^T @ b[M,K]; optionally dbias[N] (+)= column sums of a."""
    for t, n in ((a, "a"), (b, "b")):
        if not t.is_cuda or t.dtype != bf16 or t.dim() != 2 or t.stride(1) != 1:
            raise _lib.VitkError(f"gemm_tn: {n} must be a 2-D bf16 GPU tensor with contiguous rows")
    M, N = a.shape
    Mb, K = b.shape
    if M != Mb:
        raise _lib.VitkError(f"gemm_tn: M mismatch {M} vs {Mb}")
    if out.dtype != f32 or tuple(out.shape) != (N, K) or out.stride(1) != 1:
        raise _lib.VitkError(f"gemm_tn: out must be fp32 [{N},{K}] with contiguous rows")
    lib = _lib.load()
    need = lib.vitk_gemm_tn_workspace(M, N, K)
    w = (ws or _default_ws).get(need, a.device)
    check(lib.vitk_gemm_tn(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K, _ptr(dbias),
                           int(accumulate), _ptr(w), w.numel(), _stream()), "gemm_tn")
    return out


# ------------------------------------------------------------------------------------------------ attention
def _head_dim(qkv, B, N, H):
    if qkv.dim() != 2 or qkv.shape[0] != B * N or qkv.shape[1] % (3 * H):
        raise _lib.VitkError(f"attention: qkv must be [B*N, 3*H*head_dim] = [{B * N}, 3*{H}*d], got {tuple(qkv.shape)}")
    return qkv.shape[1] // (3 * H)


def attention_fwd(qkv, B, N, H, scale):
    """qkv bf16 [B*N, 3*H*d] -> (out bf16 [B*N, H*d], lse fp32 [B,H,N]).  d = 64: the MFMA kernels (<= 608 tokens); d = 96 up to
    224 tokens: the MFMA kernels with 256-byte LDS rows; any other multiple of 8 up to 96: the vector-arithmetic kernels
    (<= 256 tokens)."""
    _chk(qkv, bf16, "qkv")
    d = _head_dim(qkv, B, N, H)
    out = torch.empty((B * N, H * d), dtype=bf16, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=f32, device=qkv.device)
    if d == 64:
        check(_lib.load().vitk_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, N, H, scale, _stream()), "attention_fwd")
    else:
        check(_lib.load().vitk_attention_generic_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, N, H, d, scale, _stream()), "attention_generic_fwd")
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, N, H, scale, ws: Workspace = None, dqkv=None):
    _chk(qkv, bf16, "qkv"); _chk(out, bf16, "out"); _chk(dout, bf16, "dout"); _chk(lse, f32, "lse")
    d = _head_dim(qkv, B, N, H)
    lib = _lib.load()
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if d != 64:
        check(lib.vitk_attention_generic_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, N, H, d, scale, _stream()),
              "attention_generic_bwd")
        return dqkv
    need = lib.vitk_attention_bwd_workspace(B, N, H)
    w = (ws or _default_ws).get(need, qkv.device) if need else None
    check(lib.vitk_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, N, H, scale, _ptr(w),
                                 w.numel() if w is not None else 0, _stream()), "attention_bwd")
    return dqkv


# ------------------------------------------------------------------------------------- embedding / pooling
def im2col_patches(img, p):
    _chk(img, f32, "img")
    B, Cc, Hh, Ww = img.shape
    out = torch.empty((B * (Hh // p) * (Ww // p), Cc * p * p), dtype=bf16, device=img.device)
    check(_lib.load().vitk_im2col_patches(_ptr(img), _ptr(out), B, Cc, Hh, Ww, p, _stream()), "im2col_patches")
    return out


def prefix_tokens_fwd(tokens, pos, x, B, N, D, prefix):
    _chk(tokens, f32, "tokens"); _chk(pos, f32, "pos"); _chk(x, bf16, "x")
    check(_lib.load().vitk_prefix_tokens_fwd(_ptr(tokens), _ptr(pos), _ptr(x), B, N, D, prefix, _stream()), "prefix_tokens_fwd")


def pos_tokens_bwd(dx, dpos, dtokens, B, N, D, prefix, accumulate=False):
    _chk(dx, bf16, "dx"); _chk(dpos, f32, "dpos")
    check(_lib.load().vitk_pos_tokens_bwd(_ptr(dx), _ptr(dpos), _ptr(dtokens), int(accumulate), B, N, D, prefix, _stream()),
          "pos_tokens_bwd")


def pool_avg_fwd(x, B, N, D, first):
    _chk(x, bf16, "x")
    y = torch.empty((B, D), dtype=bf16, device=x.device)
    check(_lib.load().vitk_pool_avg_fwd(_ptr(x), _ptr(y), B, N, D, first, _stream()), "pool_avg_fwd")
    return y


def pool_avg_bwd(dy, B, N, D, first, dx=None):
    _chk(dy, bf16, "dy")
    if dx is None:
        dx = torch.empty((B * N, D), dtype=bf16, device=dy.device)
    check(_lib.load().vitk_pool_avg_bwd(_ptr(dy), _ptr(dx), B, N, D, first, _stream()), "pool_avg_bwd")
    return dx


def gather_rows(x, B, N, D, token):
    _chk(x, bf16, "x")
    y = torch.empty((B, D), dtype=bf16, device=x.device)
    check(_lib.load().vitk_gather_rows(_ptr(x), _ptr(y), B, N, D, token, _stream()), "gather_rows")
    return y


def scatter_rows(dy, dx, B, N, D, token, zero_others):
    _chk(dy, bf16, "dy"); _chk(dx, bf16, "dx")
    check(_lib.load().vitk_scatter_rows(_ptr(dy), _ptr(dx), B, N, D, token, int(zero_others), _stream()), "scatter_rows")
    return dx


def slice_tokens(x, B, N, D, first):
    _chk(x, bf16, "x")
    y = torch.empty((B * (N - first), D), dtype=bf16, device=x.device)
    check(_lib.load().vitk_slice_tokens(_ptr(x), _ptr(y), B, N, D, first, _stream()), "slice_tokens")
    return y


# ------------------------------------------------------------------------------------------- elementwise
def rowscale(x, scale, rows_per_group, out=None):
    _chk(x, bf16, "x"); _chk(scale, f32, "scale")
    M, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().vitk_rowscale(_ptr(x), _ptr(scale), rows_per_group, _ptr(out), M, D, _stream()), "rowscale")
    return out


def rowscale_gather(x, scale, rows_per_group, smap):
    """compact scaled copy: y[c*rpg + t] = scale[smap[c]] * x[smap[c]*rpg + t]"""
    _chk(x, bf16, "x"); _chk(scale, f32, "scale"); _chk(smap, torch.int32, "smap")
    nk, D = smap.numel(), x.shape[1]
    out = torch.empty((nk * rows_per_group, D), dtype=bf16, device=x.device)
    check(_lib.load().vitk_rowscale_gather(_ptr(x), _ptr(scale), rows_per_group, _ptr(smap), nk, _ptr(out), D, _stream()), "rowscale_gather")
    return out


def copy_samples(src, dst, sample_list, elems_per_sample):
    """dst[s] = src[s] for the whole samples in sample_list (int32)"""
    _chk(src, bf16, "src"); _chk(dst, bf16, "dst"); _chk(sample_list, torch.int32, "sample_list")
    if sample_list.numel():
        check(_lib.load().vitk_copy_samples(_ptr(src), _ptr(dst), _ptr(sample_list), sample_list.numel(), int(elems_per_sample), _stream()),
              "copy_samples")
    return dst


def copy_from_pinned(src, device, out=None):
    """device copy of a pinned host tensor, made by a kernel that reads the host memory (no hipMemcpyAsync call); `out`: an
    existing device tensor of the same size to copy into"""
    if src.is_cuda or not src.is_pinned() or not src.is_contiguous():
        raise _lib.VitkError("copy_from_pinned: expected a contiguous pinned host tensor")
    if out is not None:
        if not out.is_cuda or not out.is_contiguous() or out.numel() * out.element_size() != src.numel() * src.element_size():
            raise _lib.VitkError("copy_from_pinned: out must be a contiguous device tensor of the source's size")
    dst = out if out is not None else torch.empty(src.shape, dtype=src.dtype, device=device)
    check(_lib.load().vitk_copy_from_host(src.data_ptr(), _ptr(dst), src.numel() * src.element_size(), _stream()), "copy_from_host")
    return dst


def cast_f32_to_bf16(src, dst=None):
    _chk(src, f32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=bf16, device=src.device)
    check(_lib.load().vitk_cast_f32_to_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()), "cast_f32_to_bf16")
    return dst


def cast_bf16_to_f32(src, dst=None):
    _chk(src, bf16, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=f32, device=src.device)
    check(_lib.load().vitk_cast_bf16_to_f32(_ptr(src), _ptr(dst), src.numel(), _stream()), "cast_bf16_to_f32")
    return dst


def transpose_bf16(src, dst=None):
    _chk(src, bf16, "src")
    rows, cols = src.shape
    if dst is None:
        dst = torch.empty((cols, rows), dtype=bf16, device=src.device)
    check(_lib.load().vitk_transpose_bf16(_ptr(src), _ptr(dst), rows, cols, _stream()), "transpose_bf16")
    return dst


def transpose_plan(pairs):
    """Device descriptor table for transpose_bf16_batched: pairs = [(src [r,c] bf16, dst [c,r] bf16), ...]."""
    rec, first = [], 0
    for src, dst in pairs:
        _chk(src, bf16, "src"); _chk(dst, bf16, "dst")
        r, c = src.shape
        if tuple(dst.shape) != (c, r) or not src.is_contiguous() or not dst.is_contiguous():
            raise _lib.VitkError("transpose_plan: dst must be the contiguous [cols, rows] partner of a contiguous src")
        rec.append([src.data_ptr(), dst.data_ptr(), r, c, first])
        first += ((r + 63) // 64) * ((c + 63) // 64)
    rec.append([0, 0, 0, 0, first])
    return torch.tensor(rec, dtype=torch.int64, device=pairs[0][0].device), len(pairs), first


def transpose_bf16_batched(plan):
    desc, n, total = plan
    check(_lib.load().vitk_transpose_bf16_batched(_ptr(desc), n, total, _stream()), "transpose_bf16_batched")


def add_bf16(a, b, out=None):
    _chk(a, bf16, "a"); _chk(b, bf16, "b")
    if out is None:
        out = torch.empty_like(a)
    check(_lib.load().vitk_add_bf16(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "add_bf16")
    return out


# ------------------------------------------------------------------------------------------------- losses
def _logits2d(x, name):
    if not x.is_cuda or x.dtype != f32 or x.dim() != 2 or x.stride(1) != 1:
        raise _lib.VitkError(f"{name}: expected fp32 GPU logits [B,C] with contiguous rows")


def xent_label(logits, target, smoothing, grad_scale, want_grad=True):
    """-> (loss_rows fp32 [B], dlogits fp32 [B,C] or None)"""
    _logits2d(logits, "xent_label")
    _chk(target, torch.int64, "target")
    Bn, Cn = logits.shape
    loss_rows = torch.empty(Bn, dtype=f32, device=logits.device)
    dl = torch.empty((Bn, Cn), dtype=f32, device=logits.device) if want_grad else None
    check(_lib.load().vitk_xent_label_fwd_bwd(_ptr(logits), logits.stride(0), _ptr(target), float(smoothing), _ptr(loss_rows),
                                              _ptr(dl), Cn, float(grad_scale), Bn, Cn, _stream()), "xent_label")
    return loss_rows, dl


def xent_soft(logits, target, grad_scale, want_grad=True):
    _logits2d(logits, "xent_soft")
    _logits2d(target, "xent_soft target")
    Bn, Cn = logits.shape
    loss_rows = torch.empty(Bn, dtype=f32, device=logits.device)
    dl = torch.empty((Bn, Cn), dtype=f32, device=logits.device) if want_grad else None
    check(_lib.load().vitk_xent_soft_fwd_bwd(_ptr(logits), logits.stride(0), _ptr(target), target.stride(0), _ptr(loss_rows),
                                             _ptr(dl), Cn, float(grad_scale), Bn, Cn, _stream()), "xent_soft")
    return loss_rows, dl


def kd_kl(student, teacher, temperature, grad_scale, want_grad=True):
    _logits2d(student, "kd_kl student")
    _logits2d(teacher, "kd_kl teacher")
    Bn, Cn = student.shape
    loss_rows = torch.empty(Bn, dtype=f32, device=student.device)
    ds = torch.empty((Bn, Cn), dtype=f32, device=student.device) if want_grad else None
    check(_lib.load().vitk_kd_kl_fwd_bwd(_ptr(student), student.stride(0), _ptr(teacher), teacher.stride(0), float(temperature),
                                         _ptr(loss_rows), _ptr(ds), Cn, float(grad_scale), Bn, Cn, _stream()), "kd_kl")
    return loss_rows, ds


def argmax_rows(logits):
    _logits2d(logits, "argmax_rows")
    Bn, Cn = logits.shape
    idx = torch.empty(Bn, dtype=torch.int64, device=logits.device)
    check(_lib.load().vitk_argmax_rows(_ptr(logits), logits.stride(0), _ptr(idx), Bn, Cn, _stream()), "argmax_rows")
    return idx


# ----------------------------------------------------------------------------------------------- optimizer
def adamw_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, p_bf16=None, ema=None, ema_decay=0.0):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, f32, n)
    check(_lib.load().vitk_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16), _ptr(ema), float(ema_decay), p.numel(),
                                      float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                      float(grad_scale), _stream()), "adamw_step")


def sumsq(g, out, accumulate=False, ws: Workspace = None):
    _chk(g, f32, "g"); _chk(out, f32, "out")
    lib = _lib.load()
    w = (ws or _default_ws).get(lib.vitk_sumsq_workspace(g.numel()), g.device)
    check(lib.vitk_sumsq(_ptr(g), g.numel(), _ptr(out), int(accumulate), _ptr(w), w.numel(), _stream()), "sumsq")
    return out


def ema_update(ema, p, decay):
    _chk(ema, f32, "ema"); _chk(p, f32, "p")
    check(_lib.load().vitk_ema_update(_ptr(ema), _ptr(p), float(decay), p.numel(), _stream()), "ema_update")


def clip_grad_scale(g, sumsq_t, max_norm):
    _chk(g, f32, "g"); _chk(sumsq_t, f32, "sumsq")
    check(_lib.load().vitk_clip_grad_scale(_ptr(g), g.numel(), _ptr(sumsq_t), float(max_norm), _stream()), "clip_grad_scale")


_prof_on = False


def prof_enable(on: bool):
    global _prof_on
    check(_lib.load().vitk_prof_enable(int(on)), "prof_enable")
    _prof_on = bool(on)


def prof_enabled() -> bool:
    return _prof_on


PROF_ALL, PROF_GEMM_NT, PROF_GEMM_TN, PROF_LN_FWD, PROF_LN_BWD, PROF_ATTN_FWD, PROF_ATTN_BWD, PROF_ADAMW = -1, 0, 1, 2, 3, 4, 5, 6


def prof_read(kind=PROF_ALL):
    """(total ms, total work, launches) of the event-bracketed launches of one kind; work = flop for the GEMM kinds, bytes else"""
    import ctypes as C
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    check(_lib.load().vitk_prof_read(kind, C.byref(ms), C.byref(fl), C.byref(n)), "prof_read")
    return ms.value, fl.value, n.value


def prof_reset():
    check(_lib.load().vitk_prof_reset(), "prof_reset")


# ------------------------------------------------------------------------------------------ mixup / cutmix
def mixup_batch_(x, lam: float):
    """in place: x = x*lam + x.flip(0)*(1-lam) on an fp32 [B, ...] batch"""
    _chk(x, f32, "x")
    B = x.shape[0]
    check(_lib.load().vitk_mixup_batch(_ptr(x), B, x.numel() // B, float(lam), float(1. - lam), _stream()), "mixup_batch")
    return x


def cutmix_batch_(x, yl: int, yh: int, xl: int, xh: int):
    """in place: x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh] on an fp32 [B,C,H,W] batch"""
    _chk(x, f32, "x")
    B, Cc, Hh, Ww = x.shape
    check(_lib.load().vitk_cutmix_batch(_ptr(x), B, Cc, Hh, Ww, int(yl), int(yh), int(xl), int(xh), _stream()), "cutmix_batch")
    return x


def mixup_targets(target, num_classes: int, lam: float, on_value: float, off_value: float):
    """int64 labels [B] -> soft targets fp32 [B, num_classes] mixed with the flipped batch's"""
    _chk(target, torch.int64, "target")
    B = target.shape[0]
    out = torch.empty((B, num_classes), dtype=f32, device=target.device)
    check(_lib.load().vitk_mixup_targets(_ptr(target), _ptr(out), num_classes, B, num_classes, float(lam), float(1. - lam),
                                         float(on_value), float(off_value), _stream()), "mixup_targets")
    return out


MIXUP_SAMPLE = np.dtype([("kind", "<i4"), ("lam", "<f4"), ("one_minus_lam", "<f4"), ("yl", "<i4"), ("yh", "<i4"), ("xl", "<i4"), ("xh", "<i4"),
                         ("reserved", "<i4")])      # struct vitk_mixup_sample


def mixup_elem_(x, params, params_dev=None):
    """in place on an fp32 [B,C,H,W] batch: sample b mixed with the original of sample B-1-b as params[b] says (MIXUP_SAMPLE
    records: kind 0 none / 1 mixup / 2 cutmix, lam, one_minus_lam, box) - timm's Mixup modes 'elem' and 'pair'"""
    _chk(x, f32, "x")
    if x.dim() != 4:
        raise _lib.VitkError(f"x: expected [B, C, H, W], got {tuple(x.shape)}")
    B, Cc, Hh, Ww = x.shape
    if params.dtype != MIXUP_SAMPLE or params.shape != (B,) or not params.flags.c_contiguous:
        raise _lib.VitkError(f"params: expected {B} contiguous MIXUP_SAMPLE records")
    if params_dev is None:
        params_dev = torch.from_numpy(params.view(np.uint8).reshape(-1).copy()).to(x.device)
    check(_lib.load().vitk_mixup_elem(_ptr(x), B, Cc, Hh, Ww, params.ctypes.data, _ptr(params_dev), _stream()), "mixup_elem")
    return x


def mixup_targets_elem(target, num_classes: int, lam_rows, on_value: float, off_value: float):
    """int64 labels [B] + one lambda per sample (fp32 [B] on the device) -> soft targets fp32 [B, num_classes]"""
    _chk(target, torch.int64, "target")
    _chk(lam_rows, f32, "lam_rows")
    B = target.shape[0]
    if lam_rows.numel() != B:
        raise _lib.VitkError(f"lam_rows: expected {B} values, got {lam_rows.numel()}")
    out = torch.empty((B, num_classes), dtype=f32, device=target.device)
    check(_lib.load().vitk_mixup_targets_elem(_ptr(target), _ptr(out), num_classes, B, num_classes, _ptr(lam_rows), float(on_value),
                                              float(off_value), _stream()), "mixup_targets_elem")
    return out


# ------------------------------------------------------------------------------------------ input pipeline tail
def preprocess_images(src, flip=None, out=None, window=None, value_range=(0.0, 1.0), mean=None, std=None, patch=None):
    """src uint8 / fp32 [B, Hs, Ws, 3] on the GPU -> fp32 NCHW [B, 3, H, W]: flip_lr for flip[b] != 0 (uint8 [B]), value_range,
    HWC -> CHW, (v - mean) / std; window = (top, left, H, W) crops every image (default: the whole image).
    patch = p: the result is the model's bf16 patch matrix [B * (H/p) * (W/p), 3 * p * p] instead (what the patch-embedding GEMM
    reads; bit-identical to im2col_patches of the fp32 batch, without that batch being written and read in between)."""
    import ctypes
    if patch is not None:
        return _preprocess_patches(src, flip, out, window, value_range, mean, std, int(patch))
    if src.dtype not in (torch.uint8, f32):
        raise _lib.VitkError(f"src: expected uint8 or float32 images, got {src.dtype}")
    _chk(src, src.dtype, "src")
    if src.dim() != 4 or src.shape[3] != 3:
        raise _lib.VitkError(f"src: expected [B, H, W, 3], got {tuple(src.shape)}")
    B, Hs, Ws, _ = src.shape
    top, left, H, W = window if window is not None else (0, 0, Hs, Ws)
    if flip is not None:
        _chk(flip, torch.uint8, "flip")
        if flip.numel() != B:
            raise _lib.VitkError(f"flip: expected {B} flags, got {flip.numel()}")
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=f32, device=src.device)
    else:
        _chk(out, f32, "out")
        if tuple(out.shape) != (B, 3, H, W):
            raise _lib.VitkError(f"out: expected {(B, 3, H, W)}, got {tuple(out.shape)}")
    if (mean is None) != (std is None):
        raise _lib.VitkError("mean and std go together")
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    check(_lib.load().vitk_preprocess_images(_ptr(src), int(src.dtype == torch.uint8), _ptr(flip), _ptr(out), B, int(H), int(W), Hs, Ws,
                                             int(top), int(left), float(value_range[0]), float(value_range[1]), m, s, _stream()),
          "preprocess_images")
    return out


def _preprocess_patches(src, flip, out, window, value_range, mean, std, patch):
    import ctypes
    if src.dtype not in (torch.uint8, f32):
        raise _lib.VitkError(f"src: expected uint8 or float32 images, got {src.dtype}")
    _chk(src, src.dtype, "src")
    if src.dim() != 4 or src.shape[3] != 3:
        raise _lib.VitkError(f"src: expected [B, H, W, 3], got {tuple(src.shape)}")
    B, Hs, Ws, _ = src.shape
    top, left, H, W = window if window is not None else (0, 0, Hs, Ws)
    if patch <= 0 or patch % 8 or H % patch or W % patch:
        raise _lib.VitkError(f"patch size {patch} must be a multiple of 8 that divides the {H} x {W} window")
    if flip is not None:
        _chk(flip, torch.uint8, "flip")
        if flip.numel() != B:
            raise _lib.VitkError(f"flip: expected {B} flags, got {flip.numel()}")
    shape = (B * (H // patch) * (W // patch), 3 * patch * patch)
    if out is None:
        out = torch.empty(shape, dtype=bf16, device=src.device)
    else:
        _chk(out, bf16, "out")
        if tuple(out.shape) != shape:
            raise _lib.VitkError(f"out: expected {shape}, got {tuple(out.shape)}")
    if (mean is None) != (std is None):
        raise _lib.VitkError("mean and std go together")
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    check(_lib.load().vitk_preprocess_patches(_ptr(src), int(src.dtype == torch.uint8), _ptr(flip), _ptr(out), patch, B, int(H), int(W), Hs, Ws,
                                              int(top), int(left), float(value_range[0]), float(value_range[1]), m, s, _stream()),
          "preprocess_patches")
    return out


RESAMPLE_DESC = np.dtype([("src_offset", "<i8"), ("src_h", "<i4"), ("src_w", "<i4"), ("box_y", "<i4"), ("box_x", "<i4"), ("box_h", "<i4"),
                          ("box_w", "<i4"), ("out_h", "<i4"), ("out_w", "<i4"), ("win_y", "<i4"), ("win_x", "<i4"), ("flip", "<i4"),
                          ("reserved", "<i4")])          # vitk_resample_desc (include/vitk.h), 56 bytes


def resample_images(src, desc, size, kernel="triangle", out=None, value_range=(0.0, 1.0), mean=None, std=None, desc_dev=None, patch=None):
    """Decoded uint8 images of different sizes, packed one after the other in `src` (1-D uint8 on the GPU), -> fp32 NCHW
    [B, 3, size, size]: per image the box desc['box_*'] is resized to desc['out_*'] with tf.image.resize(antialias=True)'s
    arithmetic ('triangle' = BILINEAR, 'keyscubic' = BICUBIC) and the size x size window at desc['win_*'] of the result is
    kept; then flip / value_range / normalisation as preprocess_images.  desc: numpy structured array of RESAMPLE_DESC (host);
    desc_dev: the same bytes already on the GPU (uint8 tensor), copied here when not given."""
    import ctypes
    _chk(src, torch.uint8, "src")
    desc = np.ascontiguousarray(desc, dtype=RESAMPLE_DESC)
    B = int(desc.shape[0])
    if kernel not in ("triangle", "keyscubic"):
        raise _lib.VitkError(f"resample_images: kernel must be 'triangle' or 'keyscubic', got {kernel!r}")
    if desc_dev is None:
        desc_dev = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(src.device)
    _chk(desc_dev, torch.uint8, "desc_dev")
    if desc_dev.numel() != B * RESAMPLE_DESC.itemsize:
        raise _lib.VitkError(f"desc_dev: expected {B * RESAMPLE_DESC.itemsize} bytes, got {desc_dev.numel()}")
    if (mean is None) != (std is None):
        raise _lib.VitkError("mean and std go together")
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    if patch is not None:          # the model's bf16 patch matrix instead of the fp32 NCHW batch (see preprocess_images)
        patch = int(patch)
        if patch <= 0 or patch % 8 or size % patch:
            raise _lib.VitkError(f"patch size {patch} must be a multiple of 8 that divides the image size {size}")
        shape = (B * (size // patch) ** 2, 3 * patch * patch)
        if out is None:
            out = torch.empty(shape, dtype=bf16, device=src.device)
        else:
            _chk(out, bf16, "out")
            if tuple(out.shape) != shape:
                raise _lib.VitkError(f"out: expected {shape}, got {tuple(out.shape)}")
        check(_lib.load().vitk_resample_patches(_ptr(src), int(src.numel()), desc.ctypes.data, _ptr(desc_dev), _ptr(out), patch, B, int(size),
                                                0 if kernel == "triangle" else 1, float(value_range[0]), float(value_range[1]), m, s,
                                                _stream()), "resample_patches")
        return out
    if out is None:
        out = torch.empty((B, 3, size, size), dtype=f32, device=src.device)
    else:
        _chk(out, f32, "out")
        if tuple(out.shape) != (B, 3, size, size):
            raise _lib.VitkError(f"out: expected {(B, 3, size, size)}, got {tuple(out.shape)}")
    check(_lib.load().vitk_resample_images(_ptr(src), int(src.numel()), desc.ctypes.data, _ptr(desc_dev), _ptr(out), B, int(size),
                                           0 if kernel == "triangle" else 1, float(value_range[0]), float(value_range[1]), m, s, _stream()),
          "resample_images")
    return out
