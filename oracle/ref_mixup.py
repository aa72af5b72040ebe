"""TEST INFRASTRUCTURE ONLY (CPU oracle): timm.data.Mixup in 'batch' mode restated with plain torch ops on the CPU.

timm (pinned 1.0.15, requirements.txt:14) is not vendored under /root/reference and not installed, so this follows the
published algorithm of timm/data/mixup.py (mixup_target, rand_bbox, cutmix_bbox_and_lam, Mixup._params_per_batch,
Mixup._mix_batch, Mixup.__call__) anchored on the reference's call sites: construction main.py:622-629, application
engine.py:132-133 and 259-260.  Parity unpinned: the reference holds no fixture for it.
"""
import numpy as np
import torch


def one_hot(x, num_classes, on_value=1., off_value=0.):
    x = x.long().view(-1, 1)
    return torch.full((x.size()[0], num_classes), off_value).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1., smoothing=0.0):
    off_value = smoothing / num_classes
    on_value = 1. - smoothing + off_value
    y1 = one_hot(target, num_classes, on_value=on_value, off_value=off_value)
    y2 = one_hot(target.flip(0), num_classes, on_value=on_value, off_value=off_value)
    return y1 * lam + y2 * (1. - lam)


def draw_params(mixup_alpha, cutmix_alpha, prob, switch_prob, img_shape, correct_lam=True):
    """(lam, box or None) for one batch, consuming numpy's global generator in the order timm does"""
    lam, use_cutmix = 1., False
    if np.random.rand() < prob:
        if mixup_alpha > 0. and cutmix_alpha > 0.:
            use_cutmix = np.random.rand() < switch_prob
            lam = float(np.random.beta(cutmix_alpha, cutmix_alpha) if use_cutmix else np.random.beta(mixup_alpha, mixup_alpha))
        elif mixup_alpha > 0.:
            lam = float(np.random.beta(mixup_alpha, mixup_alpha))
        else:
            use_cutmix = True
            lam = float(np.random.beta(cutmix_alpha, cutmix_alpha))
    if lam == 1. or not use_cutmix:
        return lam, None
    h, w = img_shape[-2:]
    ratio = np.sqrt(1 - lam)
    cut_h, cut_w = int(h * ratio), int(w * ratio)
    cy = np.random.randint(0, h, size=None)
    cx = np.random.randint(0, w, size=None)
    yl, yh = np.clip(cy - cut_h // 2, 0, h), np.clip(cy + cut_h // 2, 0, h)
    xl, xh = np.clip(cx - cut_w // 2, 0, w), np.clip(cx + cut_w // 2, 0, w)
    if correct_lam:
        lam = 1. - (yh - yl) * (xh - xl) / float(h * w)
    return lam, (int(yl), int(yh), int(xl), int(xh))


def apply(x, target, lam, box, num_classes, smoothing):
    """the tensor side of Mixup.__call__ on CPU tensors: returns (mixed x, soft targets); x is not modified"""
    x = x.clone()
    if lam != 1.:
        if box is not None:
            yl, yh, xl, xh = box
            x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
        else:
            x_flipped = x.flip(0).mul_(1. - lam)
            x.mul_(lam).add_(x_flipped)
    return x, mixup_target(target, num_classes, lam, smoothing)
