"""TEST INFRASTRUCTURE ONLY (CPU oracle): the per-element tail of the reference's input pipeline in numpy / torch CPU ops.

Follows, step by step, what the reference does to a decoded and cropped image on the host: flip_lr
(big_vision/pp/ops_image.py:69-82; the coin is an argument here), value_range (big_vision/pp/ops_general.py:26-43),
the float32 cast and HWC -> CHW transpose (datasets.py:301-307), `image.sub_(mean).div_(std)` with timm's ImageNet
constants (datasets.py:130-131, 249-252) and central_crop for images no smaller than the crop (ops_image.py:121-134,
tf.image.resize_with_crop_or_pad: offsets (h - size) // 2, (w - size) // 2).  TensorFlow is not installed here, so the
value_range step is restated with numpy float32 arithmetic (one IEEE operation per TF op); the normalise step runs the
reference's own torch operations.  Parity unpinned: the reference holds no fixture for its input pipeline.
"""
import numpy as np
import torch

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def value_range(image, vmin, vmax):
    scale = np.float32(vmax - vmin)
    if np.issubdtype(image.dtype, np.integer):
        image = image.astype(np.float32) / np.float32(255.0)
    else:
        image = image.astype(np.float32)
    return image * scale + np.float32(vmin)


def central_crop_offsets(h, w, size):
    return (h - size) // 2, (w - size) // 2


def preprocess_image(image_hwc, flip, vmin=0.0, vmax=1.0, normalize="imagenet", window=None):
    """one image [H, W, 3] (uint8 or float32) -> torch float32 [3, h, w]; window = (top, left, h, w) or None"""
    img = np.asarray(image_hwc)
    if window is not None:
        top, left, h, w = window
        img = img[top:top + h, left:left + w]
    if flip:
        img = img[:, ::-1]
    img = value_range(img, vmin, vmax)
    img = np.transpose(img, (2, 0, 1))
    t = torch.from_numpy(np.ascontiguousarray(img)).contiguous()
    if normalize == "imagenet":
        mean = torch.tensor(IMAGENET_DEFAULT_MEAN).view(3, 1, 1).float()
        std = torch.tensor(IMAGENET_DEFAULT_STD).view(3, 1, 1).float()
        t = t.sub_(mean).div_(std)
    return t


def preprocess_batch(images_nhwc, flips=None, **kw):
    return torch.stack([preprocess_image(images_nhwc[b], bool(flips[b]) if flips is not None else False, **kw)
                        for b in range(len(images_nhwc))])
