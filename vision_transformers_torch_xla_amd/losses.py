"""Criteria of the reference's training loop, computed by fused HIP kernels (log-softmax, loss and gradient in
one pass over fp32 logits).

Mirrors main.py:926-935 (criterion choice: timm SoftTargetCrossEntropy when mixup is on, timm
LabelSmoothingCrossEntropy(smoothing) when smoothing > 0, else nn.CrossEntropyLoss) and the inline knowledge
distillation classes main.py:836-851 (StudentWithDistillation) and main.py:939-970 (DistillationLoss), duplicated in
test_kd.py:43-88.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import VitkError


def _as_logits(x):
    if not x.is_cuda:
        raise VitkError("criterion: logits must be on the GPU; the fused loss kernels are the only implementation")
    if x.dtype != torch.float32:
        x = x.float()
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.reshape(x.shape[0], -1).contiguous()
    return x


class _FusedLoss(torch.autograd.Function):
    """mean over the batch of a per-row loss; the kernel already produced d(mean loss)/d(logits)."""

    @staticmethod
    def forward(ctx, logits, kind, arg0, arg1):
        B = logits.shape[0]
        want = logits.requires_grad and torch.is_grad_enabled()
        if kind == "label":
            rows, dl = ops.xent_label(logits, arg0, arg1, 1.0 / B, want_grad=True)
        elif kind == "soft":
            rows, dl = ops.xent_soft(logits, arg0, 1.0 / B, want_grad=True)
        else:   # "kd": arg0 = teacher logits, arg1 = temperature; 'batchmean' = sum over rows / B
            rows, dl = ops.kd_kl(logits, arg0, arg1, 1.0 / B, want_grad=True)
        ctx.dl = dl
        return rows.mean()

    @staticmethod
    def backward(ctx, go):
        dl, ctx.dl = ctx.dl, None
        return dl * go, None, None, None


class LabelSmoothingCrossEntropy(nn.Module):
    """timm.loss.LabelSmoothingCrossEntropy: mean_b[(1-eps) * nll_b + eps * mean_c(-logp_bc)]"""

    def __init__(self, smoothing: float = 0.1):
        super().__init__()
        assert smoothing < 1.0
        self.smoothing = smoothing

    def forward(self, x, target):
        return _FusedLoss.apply(_as_logits(x), "label", target.contiguous(), float(self.smoothing))


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() with integer targets (main.py:934, engine.py:341)"""

    def forward(self, x, target):
        return _FusedLoss.apply(_as_logits(x), "label", target.contiguous(), 0.0)


class SoftTargetCrossEntropy(nn.Module):
    """timm.loss.SoftTargetCrossEntropy: mean_b sum_c -t_bc * logp_bc"""

    def forward(self, x, target):
        t = target
        if t.dtype != torch.float32 or t.stride(-1) != 1:
            t = t.float().contiguous()
        return _FusedLoss.apply(_as_logits(x), "soft", t, None)


class DistillationLoss(nn.Module):
    """(1-alpha) * base(student, y) + alpha * T^2 * KL_batchmean(softmax(teacher/T) || log_softmax(student/T))
    when given (student_logits, teacher_logits); base(outputs, y) otherwise (main.py:939-970)."""

    def __init__(self, base_criterion, alpha: float = 0.7, temperature: float = 4.0):
        super().__init__()
        self.base_criterion, self.alpha, self.temperature = base_criterion, alpha, temperature

    def forward(self, outputs, targets):
        if isinstance(outputs, tuple):
            student_logits, teacher_logits = outputs
            ce = self.base_criterion(student_logits, targets)
            kd = _FusedLoss.apply(_as_logits(student_logits), "kd", _as_logits(teacher_logits.detach()), float(self.temperature))
            return (1 - self.alpha) * ce + self.alpha * kd
        return self.base_criterion(outputs, targets)


class StudentWithDistillation(nn.Module):
    """train mode: (student(x), teacher(x) under no_grad); eval mode: student(x)  (main.py:836-851)"""

    def __init__(self, student_model, teacher_model):
        super().__init__()
        self.student = student_model
        self.teacher = teacher_model

    def forward(self, x):
        if not (self.training and self.teacher is not None):
            return self.student(x)
        # The two forward passes do not depend on each other.  Where the student has a side stream (the one its weight-gradient
        # GEMMs use in backward, idle now), the teacher runs there: each model's GEMMs fill the compute units that the other's
        # last tile rounds leave empty.  The teacher's activations are allocated under that stream, so its allocator pool
        # hands them on in that stream's order; only the logits cross over, behind the event.
        side = self.student._wgrad_stream() if (x.is_cuda and hasattr(self.student, "_wgrad_stream")
                                                and os.environ.get("VITK_KD_OVERLAP", "1") != "0") else None
        if side is None:
            student_logits = self.student(x)
            with torch.no_grad():
                teacher_logits = self.teacher(x)
            return student_logits, teacher_logits
        ready = torch.cuda.Event()
        ready.record()                                   # x is complete once everything queued so far has run
        side.wait_event(ready)
        with torch.cuda.stream(side), torch.no_grad():
            teacher_logits = self.teacher(x)
            done = torch.cuda.Event()
            done.record(side)
        student_logits = self.student(x)
        torch.cuda.current_stream().wait_event(done)     # the loss reads the teacher's logits; x may be released after this
        return student_logits, teacher_logits
