"""Mixup / CutMix with the call contract of timm.data.Mixup as the reference uses it (main.py:622-629:
Mixup(mixup_alpha, cutmix_alpha, cutmix_minmax, prob, switch_prob, mode, label_smoothing, num_classes), applied as
`samples, targets = mixup_fn(samples, targets)` before the forward pass, engine.py:132-133 / 259-260), with the
pixel and label arithmetic on the GPU (libvitk) instead of on the host before the device copy.

Host side (this file): the per-batch random draws, in timm's order and from numpy's global generator like timm
(np.random.rand for mix_prob, np.random.rand for switch_prob, np.random.beta for lambda, np.random.randint for the box
centre), so a run seeded with np.random.seed(s) draws the same lambdas and boxes.  Device side: vitk_mixup_batch,
vitk_cutmix_batch, vitk_mixup_targets for mode='batch' (the reference's default, main.py:229); vitk_mixup_elem and
vitk_mixup_targets_elem for 'elem' (one lambda / kind / box per sample) and 'pair' (one per pair of samples).

Scalar types follow the numpy the reference pins (1.26.4, requirements.txt:13): there `1 - lam` and `img_h * ratio` with a
float32 lambda are float64 (both operands scalars), so the box sizes of the per-sample modes are derived in double here too.
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
        if mode not in ('batch', 'elem', 'pair'):
            raise VitkError(f"Mixup mode {mode!r} is not one of 'batch', 'elem', 'pair'")
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

    def _params_per_elem(self, batch_size):
        """timm Mixup._params_per_elem: vectorised draws - switch coins, BOTH beta vectors (np.where evaluates its two branches),
        then the mix_prob coins"""
        lam = np.ones(batch_size, dtype=np.float32)
        use_cutmix = np.zeros(batch_size, dtype=bool)
        if self.mixup_enabled:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand(batch_size) < self.switch_prob
                lam_mix = np.where(use_cutmix, np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size),
                                   np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size))
            elif self.mixup_alpha > 0.:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size)
            elif self.cutmix_alpha > 0.:
                use_cutmix = np.ones(batch_size, dtype=bool)
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = np.where(np.random.rand(batch_size) < self.mix_prob, lam_mix.astype(np.float32), lam)
        return lam, use_cutmix

    def _sample_plan(self, shape):
        """Host side of Mixup._mix_elem / _mix_pair: (MIXUP_SAMPLE records for vitk_mixup_elem, float32 lambda per sample for the
        targets).  Boxes are drawn in sample order, one per cutmix sample (elem) or pair (pair), as timm's loops draw them."""
        B = shape[0]
        n = B if self.mode == 'elem' else B // 2
        lam_batch, use_cutmix = self._params_per_elem(n)
        prm = np.zeros(B, dtype=ops.MIXUP_SAMPLE)
        for i in range(n):
            lam = lam_batch[i]
            if lam == 1.:
                continue
            if use_cutmix[i]:
                (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(shape[1:], float(lam), ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
                prm[i] = (2, 0., 0., int(yl), int(yh), int(xl), int(xh), 0)
                lam_batch[i] = lam
            else:
                prm[i] = (1, lam, np.float32(1. - float(lam)), 0, 0, 0, 0, 0)
            if self.mode == 'pair':
                prm[B - 1 - i] = prm[i]
        if self.mode == 'pair':
            lam_batch = np.concatenate((lam_batch, lam_batch[::-1]))
        return prm, lam_batch.astype(np.float32)

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
        if x.dim() != 4 or x.shape[0] != target.shape[0]:
            raise VitkError(f"Mixup mixes image batches [B, C, H, W] with B labels; got samples {tuple(x.shape)} and {target.shape[0]} labels "
                            f"(a loader's patch= option yields the model's patch matrix, which cannot be mixed: leave it off with Mixup)")
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        if not x.is_cuda:           # the loader's host batch: copy first, mix on the device
            x = x.cuda(non_blocking=True)
        if not target.is_cuda:
            target = target.cuda(non_blocking=True)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        off_value = self.label_smoothing / self.num_classes
        on_value = 1. - self.label_smoothing + off_value
        if self.mode != 'batch':
            prm, lam_rows = self._sample_plan(x.shape)
            if prm["kind"].any():
                ops.mixup_elem_(x, prm)
            lam_dev = torch.from_numpy(lam_rows).to(x.device, non_blocking=True)
            return x, ops.mixup_targets_elem(target.long().contiguous(), self.num_classes, lam_dev, on_value, off_value)
        lam = self._mix_batch(x)
        target = ops.mixup_targets(target.long().contiguous(), self.num_classes, lam, on_value, off_value)
        return x, target
