"""Mixup / CutMix with the call contract of timm.data.Mixup as the reference uses it (main.py:622-629:
Mixup(mixup_alpha, cutmix_alpha, cutmix_minmax, prob, switch_prob, mode, label_smoothing, num_classes), applied as
`samples, targets = mixup_fn(samples, targets)` before the forward pass, engine.py:132-133 / 259-260), with the
pixel and label arithmetic on the GPU (libvitk) instead of on the host before the device copy.

Host side (this file): the per-batch random draws, in timm's order and from numpy's global generator like timm
(np.random.rand for mix_prob, np.random.rand for switch_prob, np.random.beta for lambda, np.random.randint for the box
centre), so a run seeded with np.random.seed(s) draws the same lambdas and boxes.  Device side: vitk_mixup_batch,
vitk_cutmix_batch, vitk_mixup_targets.  Only mode='batch' (the reference's default, main.py:229) is implemented.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import VitkError


def rand_bbox(img_shape, lam, margin=0., count=None):
    """timm.data.mixup.rand_bbox: a box of area ratio (1 - lam) around a uniformly drawn centre, clipped to the image"""
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y, size=count)
    cx = np.random.randint(0 + margin_x, img_w - margin_x, size=count)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax, count=None):
    """timm.data.mixup.rand_bbox_minmax: box sides drawn as a fraction of the image sides in [min, max]"""
    assert len(minmax) == 2
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]), size=count)
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]), size=count)
    yl = np.random.randint(0, img_h - cut_h, size=count)
    xl = np.random.randint(0, img_w - cut_w, size=count)
    yu = yl + cut_h
    xu = xl + cut_w
    return yl, yu, xl, xu


def cutmix_bbox_and_lam(img_shape, lam, ratio_minmax=None, correct_lam=True, count=None):
    if ratio_minmax is not None:
        yl, yu, xl, xu = rand_bbox_minmax(img_shape, ratio_minmax, count=count)
    else:
        yl, yu, xl, xu = rand_bbox(img_shape, lam, count=count)
    if correct_lam or ratio_minmax is not None:
        bbox_area = (yu - yl) * (xu - xl)
        lam = 1. - bbox_area / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


class Mixup:
    def __init__(self, mixup_alpha=1., cutmix_alpha=0., cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha = mixup_alpha
        self.cutmix_alpha = cutmix_alpha
        self.cutmix_minmax = cutmix_minmax
        if self.cutmix_minmax is not None:
            assert len(self.cutmix_minmax) == 2
            self.cutmix_alpha = 1.0       # force cutmix alpha == 1.0 when minmax active to keep logic simple & safe
        self.mix_prob = prob
        self.switch_prob = switch_prob
        self.label_smoothing = label_smoothing
        self.num_classes = num_classes
        if mode != 'batch':
            raise VitkError(f"Mixup mode {mode!r} is not implemented (only 'batch', the reference's default)")
        self.mode = mode
        self.correct_lam = correct_lam
        self.mixup_enabled = True       # set to False to disable mixing (intended to be set by the train loop)

    def _params_per_batch(self):
        lam = 1.
        use_cutmix = False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = float(lam_mix)
        return lam, use_cutmix

    def _mix_batch(self, x):
        lam, use_cutmix = self._params_per_batch()
        if lam == 1.:
            return 1.
        if use_cutmix:
            (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam, ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
            ops.cutmix_batch_(x, int(yl), int(yh), int(xl), int(xh))
        else:
            ops.mixup_batch_(x, lam)
        return lam

    def __call__(self, x, target):
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        if not x.is_cuda:           # the loader's host batch: copy first, mix on the device
            x = x.cuda(non_blocking=True)
        if not target.is_cuda:
            target = target.cuda(non_blocking=True)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        lam = self._mix_batch(x)
        off_value = self.label_smoothing / self.num_classes
        on_value = 1. - self.label_smoothing + off_value
        target = ops.mixup_targets(target.long().contiguous(), self.num_classes, lam, on_value, off_value)
        return x, target
