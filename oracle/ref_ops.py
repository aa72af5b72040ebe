"""Per-op CPU restatements (test oracle; see oracle/__init__.py).

Every function computes in fp32 (or fp64 when given doubles) with plain torch ops.  ``emulate=True`` additionally
rounds to bf16 at exactly the tensor boundaries where the HIP kernels store bf16, so that the kernels can be
compared at 1e-3 relative; ``emulate=False`` is the reference's fp32 semantics.

Layer arithmetic follows timm 1.0.x as used by the reference:
  LayerNorm  F.layer_norm(x,(D,),w,b,eps=1e-6)                 models/vision_transformer.py:148,163,530,616
  Attention  qkv Linear -> [B,N,3,H,d].permute(2,0,3,1,4) -> softmax(q k^T d^-.5) v -> proj Linear   :149-159
  Mlp        fc2(GELU_erf(fc1 x))                                                                     :164-171
  DropPath   x * Bernoulli(keep)/keep per sample, train only                                          :161,173
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def rb(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bf16, returned in the input dtype"""
    return x.to(torch.bfloat16).to(x.dtype)


class _Round(torch.autograd.Function):
    """Identity up to bf16 rounding: rounds the value in forward and/or the gradient in backward."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rb(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


def rnd(x, emulate, fwd=True, bwd=True):
    return _Round.apply(x, fwd, bwd) if emulate else x


# ----------------------------------------------------------------------------------------------- layers
def layernorm(x, w, b, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def dgelu(x):
    """d/dx of the exact (erf) GELU: Phi(x) + x * pdf(x)"""
    return 0.5 * (1.0 + torch.erf(x * (0.5 ** 0.5))) + x * torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))


class _GeluKeptDerivative(torch.autograd.Function):
    """exact GELU whose backward multiplies by the derivative as the HIP path keeps it between the passes: evaluated at the
    fp32 pre-activation, then rounded to bf16 - a bf16 storage point like every other saved activation (the pre-activation
    itself is not one: nothing stores it)"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(rb(dgelu(x)))
        return F.gelu(x)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return g * d


def gelu(x, emulate=False):
    """exact erf form, nn.GELU() default (timm Mlp act_layer); emulate=True: see _GeluKeptDerivative"""
    return _GeluKeptDerivative.apply(x) if emulate else F.gelu(x)


def im2col_patches(img: torch.Tensor, p: int) -> torch.Tensor:
    """[B,C,H,W] -> [B, gh*gw, C*p*p] with k = c*p*p + ky*p + kx and patch index gy*gw + gx, i.e. the operand order of
    Conv2d(C,D,k=p,s=p) + flatten(2).transpose(1,2) (timm PatchEmbed, models/vision_transformer.py:552-560,936)."""
    B, C, H, W = img.shape
    gh, gw = H // p, W // p
    x = img.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * p * p)


class _Attention(torch.autograd.Function):
    """softmax(q k^T * scale) v over heads of dim d, explicit forward/backward.

    With emulate=True the bf16 roundings of the HIP kernels are reproduced: P is rounded before P@V and before
    P^T@dO, dS / scale is rounded before its products with K and Q (their sums are scaled in fp32), O is rounded on output;
    everything else is fp32."""

    @staticmethod
    def forward(ctx, qkv, H, scale, emulate):
        B, N, D3 = qkv.shape
        d = D3 // (3 * H)
        q, k, v = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)   # [B,H,N,d]
        s = (q @ k.transpose(-1, -2)) * scale
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        pb = rb(p) if emulate else p
        o = (pb @ v) / l
        if emulate:
            o = rb(o)
        lse = m + torch.log(l)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.emulate, ctx.H = scale, emulate, H
        return o.transpose(1, 2).reshape(B, N, H * d)

    @staticmethod
    def backward(ctx, do_):
        q, k, v, o, lse = ctx.saved_tensors
        return _attention_backward(q, k, v, o, lse, do_, ctx.scale, ctx.emulate), None, None, None


def _attention_backward(q, k, v, o, lse, do_, scale, emulate):
    """q, k, v, o [B,H,N,d], lse [B,H,N,1], do_ [B,N,H*d] -> dqkv [B,N,3*H*d]"""
    B, H, N, d = q.shape
    do = do_.reshape(B, N, H, d).transpose(1, 2)
    p = torch.exp((q @ k.transpose(-1, -2)) * scale - lse)
    delta = (do * o).sum(dim=-1, keepdim=True)
    pb = rb(p) if emulate else p
    dv = pb.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - delta)              # dS / scale: the kernels store this (bf16) and scale the dQ, dK sums once, in fp32
    if emulate:
        ds = rb(ds)
    dq = (ds @ k) * scale
    dk = (ds.transpose(-1, -2) @ q) * scale
    return torch.stack([dq, dk, dv], dim=0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * d)


def attention_backward(qkv, out, lse, dout, H, scale=None, emulate=False):
    """The backward of `attention` from GIVEN forward results: qkv [B,N,3*H*d], out [B,N,H*d] and lse [B,H,N] as a forward
    pass produced them, dout [B,N,H*d] -> dqkv.  (Lets a test hand the oracle the very O and log-sum-exp the kernel under
    test consumed, so that both sides start from identical inputs.)"""
    B, N, D3 = qkv.shape
    d = D3 // (3 * H)
    q, k, v = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    o = out.reshape(B, N, H, d).transpose(1, 2)
    return _attention_backward(q, k, v, o, lse.reshape(B, H, N, 1), dout, d ** -0.5 if scale is None else scale, emulate)


def attention(qkv, H, scale=None, emulate=False):
    d = qkv.shape[-1] // (3 * H)
    return _Attention.apply(qkv, H, d ** -0.5 if scale is None else scale, emulate)


# ------------------------------------------------------------------------------------------------ losses
def label_smoothing_ce(logits, target, smoothing=0.1):
    """timm LabelSmoothingCrossEntropy (main.py:930-931); smoothing=0 equals nn.CrossEntropyLoss (main.py:934)."""
    logp = F.log_softmax(logits, dim=-1)
    nll = -logp.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
    smooth = -logp.mean(dim=-1)
    return ((1.0 - smoothing) * nll + smoothing * smooth).mean()


def soft_target_ce(logits, target):
    """timm SoftTargetCrossEntropy (main.py:928)."""
    return torch.sum(-target * F.log_softmax(logits, dim=-1), dim=-1).mean()


def distillation_loss(student_logits, teacher_logits, targets, base, alpha=0.7, temperature=4.0):
    """DistillationLoss (main.py:939-970, test_kd.py:43-88)."""
    ce = base(student_logits, targets)
    s = torch.log_softmax(student_logits / temperature, dim=1)
    t = torch.softmax(teacher_logits / temperature, dim=1)
    kd = F.kl_div(s, t, reduction="batchmean") * (temperature ** 2)
    return (1 - alpha) * ce + alpha * kd


def accuracy_topk(output, target, topk=(1,)):
    """timm.utils.accuracy (engine.py:405): percentage of samples whose target is in the top-k."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.shape[0] for k in topk]


# -------------------------------------------------------------------------------------- schedules / optim
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """utils/__init__.py:667-684."""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / (len(iters)))) for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def parameter_groups(model, weight_decay, skip_list=()):
    """optim_factory.py:151-195 (non-TPU branch, no layer decay): returns [decay group, no_decay group] in first-seen
    order, each {'weight_decay', 'params', 'lr_scale', 'names'}."""
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        no_decay = len(param.shape) == 1 or name.endswith(".bias") or name in skip_list
        gname = "no_decay" if no_decay else "decay"
        if gname not in groups:
            groups[gname] = {"weight_decay": 0.0 if no_decay else weight_decay, "params": [], "lr_scale": 1.0, "names": []}
        groups[gname]["params"].append(param)
        groups[gname]["names"].append(name)
    return list(groups.values())


def adamw_reference_step(p, g, m, v, lr, beta1, beta2, eps, wd, step):
    """torch.optim.AdamW single-tensor arithmetic (the optimizer optim_factory.py:248-249 builds), out of place."""
    p = p * (1 - lr * wd)
    m = torch.lerp(m, g, 1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
