"""TEST INFRASTRUCTURE ONLY (CPU oracle): timm.data.Mixup (modes 'batch', 'elem', 'pair') restated with plain torch ops on the CPU.

timm (pinned 1.0.15, requirements.txt:14) is not vendored under /root/reference and not installed, so this follows the
published algorithm of timm/data/mixup.py (mixup_target, rand_bbox, cutmix_bbox_and_lam, Mixup._params_per_batch,
Mixup._mix_batch, Mixup._params_per_elem, Mixup._mix_elem, Mixup._mix_pair, Mixup.__call__) anchored on the reference's call sites: construction main.py:622-629, application
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


def _box(img_hw, lam, correct_lam=True):
    """rand_bbox + the lambda correction for ONE sample; with the reference's numpy (1.26.4) `1 - lam` of a float32 scalar is
    float64, so the ratio and the box sizes are formed in double"""
    h, w = img_hw
    ratio = np.sqrt(1 - float(lam))
    cut_h, cut_w = int(h * ratio), int(w * ratio)
    cy = np.random.randint(0, h, size=None)
    cx = np.random.randint(0, w, size=None)
    yl, yh = np.clip(cy - cut_h // 2, 0, h), np.clip(cy + cut_h // 2, 0, h)
    xl, xh = np.clip(cx - cut_w // 2, 0, w), np.clip(cx + cut_w // 2, 0, w)
    if correct_lam:
        lam = 1. - (yh - yl) * (xh - xl) / float(h * w)
    return (int(yl), int(yh), int(xl), int(xh)), lam


def draw_params_per_elem(mixup_alpha, cutmix_alpha, prob, switch_prob, n):
    """Mixup._params_per_elem: (float32 lambdas [n], use_cutmix [n]); np.where draws both beta vectors"""
    lam = np.ones(n, dtype=np.float32)
    use_cutmix = np.zeros(n, dtype=bool)
    if mixup_alpha > 0. and cutmix_alpha > 0.:
        use_cutmix = np.random.rand(n) < switch_prob
        lam_mix = np.where(use_cutmix, np.random.beta(cutmix_alpha, cutmix_alpha, size=n), np.random.beta(mixup_alpha, mixup_alpha, size=n))
    elif mixup_alpha > 0.:
        lam_mix = np.random.beta(mixup_alpha, mixup_alpha, size=n)
    else:
        use_cutmix = np.ones(n, dtype=bool)
        lam_mix = np.random.beta(cutmix_alpha, cutmix_alpha, size=n)
    lam = np.where(np.random.rand(n) < prob, lam_mix.astype(np.float32), lam)
    return lam, use_cutmix


def mix_per_sample(x, target, mode, mixup_alpha, cutmix_alpha, prob, switch_prob, num_classes, smoothing):
    """Mixup.__call__ in mode 'elem' or 'pair' on CPU tensors, written as timm's loops over a clone of the batch: returns (mixed x,
    soft targets, lambdas); consumes numpy's global generator in timm's order; x is not modified"""
    x = x.clone()
    B = len(x)
    n = B if mode == 'elem' else B // 2
    lam_batch, use_cutmix = draw_params_per_elem(mixup_alpha, cutmix_alpha, prob, switch_prob, n)
    x_orig = x.clone()
    for i in range(n):
        j = B - i - 1
        lam = lam_batch[i]
        if lam != 1.:
            if use_cutmix[i]:
                (yl, yh, xl, xh), lam = _box(x.shape[-2:], lam)
                x[i][:, yl:yh, xl:xh] = x_orig[j][:, yl:yh, xl:xh]
                if mode == 'pair':
                    x[j][:, yl:yh, xl:xh] = x_orig[i][:, yl:yh, xl:xh]
                lam_batch[i] = lam
            else:
                x[i] = x[i] * float(lam) + x_orig[j] * (1 - float(lam))
                if mode == 'pair':
                    x[j] = x[j] * float(lam) + x_orig[i] * (1 - float(lam))
    if mode == 'pair':
        lam_batch = np.concatenate((lam_batch, lam_batch[::-1]))
    lam_t = torch.tensor(lam_batch, dtype=x.dtype).unsqueeze(1)
    return x, mixup_target(target, num_classes, lam_t, smoothing), lam_batch
