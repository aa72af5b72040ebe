"""TEST INFRASTRUCTURE ONLY (CPU oracle): the crop / resize stage of the reference's input pipeline in numpy float32.

The reference runs this stage on the host inside tf.data (big_vision/pp/ops_image.py):
  train  decode_jpeg_and_inception_crop(224)   :36-66   uint8 -> float32 * (1/255); tf.image.sample_distorted_bounding_box with no
                                                         boxes (area 8-100 %, aspect 3/4-4/3, 10 attempts, else the whole image);
                                                         tf.slice; tf.image.resize(BILINEAR, antialias=True) to 224 x 224
  eval   decode | resize_small(256) | central_crop(224)  :21-33, :85-119, :121-134   float32 * (1/255); the shorter side to 256 with
                                                         tf.image.resize(BICUBIC, antialias=True), the sizes from tf.round;
                                                         tf.image.resize_with_crop_or_pad
then flip_lr / value_range / transpose / normalise (oracle/ref_preprocess.py).

TensorFlow is not installed here, so tf.image.resize(..., antialias=True) is restated from its published algorithm, the
ScaleAndTranslate op (tensorflow/core/kernels/image/scale_and_translate_op.cc at the version the reference pins, tensorflow-cpu 2.x):
  scale = out / in (float32);  inv_scale = float32(1.0 / scale);  kernel_scale = max(inv_scale, 1) when antialiasing
  for output index x:  sample = (x + 0.5) * inv_scale;   span = [ceil(sample - R * kernel_scale - 0.5), floor(sample + R * kernel_scale - 0.5)]
                       clamped to the image;  weight(src) = kernel(|src + 0.5 - sample| / kernel_scale), normalised by their sum
  kernels: 'triangle' (BILINEAR, R = 1): max(0, 1 - |x|);  'keyscubic' (BICUBIC, R = 2): Keys cubic with A = -0.5
  rows first (intermediate [out_h, in_w]), then columns; every sum starts at 0 and adds weight * value in ascending source order.
Each float operation is one IEEE float32 operation here (no fused multiply-add).  sample_distorted_bounding_box is restated from
sample_distorted_bounding_box_op.cc's GenerateRandomCrop; TensorFlow's random stream is not reproduced (boxes are an input of the
resampling functions, so tests inject them).  Parity unpinned: the reference holds no fixture for its input pipeline and neither
TensorFlow nor its JPEG decoder can run here.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


def _triangle(x):
    x = abs(x)
    return f32(1.0) - x if x < f32(1.0) else f32(0.0)


def _keys_cubic(x):
    x = abs(x)
    if x >= f32(2.0):
        return f32(0.0)
    if x >= f32(1.0):
        return ((f32(-0.5) * x + f32(2.5)) * x - f32(4.0)) * x + f32(2.0)
    return ((f32(1.5) * x - f32(2.5)) * x) * x + f32(1.0)


KERNELS = {"triangle": (_triangle, f32(1.0)), "keyscubic": (_keys_cubic, f32(2.0))}


def compute_spans(in_size: int, out_size: int, kernel: str, antialias: bool = True):
    """[(first source index, float32 weights)] for every output index (ComputeSpansCore)"""
    fn, radius = KERNELS[kernel]
    scale = f32(out_size) / f32(in_size)
    inv_scale = f32(1.0 / float(scale))
    kernel_scale = max(inv_scale, f32(1.0)) if antialias else f32(1.0)
    one_over = f32(1.0) / kernel_scale
    out = []
    for x in range(out_size):
        sample = f32(f32(x) + f32(0.5)) * inv_scale
        if sample < 0 or sample > in_size:
            out.append((0, np.zeros(0, f32)))
            continue
        start = int(math.ceil(f32(f32(sample - f32(radius * kernel_scale)) - f32(0.5))))
        end = int(math.floor(f32(f32(sample + f32(radius * kernel_scale)) - f32(0.5))))
        start = min(max(start, 0), in_size - 1)
        end = min(max(end, 0), in_size - 1) + 1
        w = np.zeros(end - start, f32)
        total = f32(0.0)
        for i, src in enumerate(range(start, end)):
            pos = f32(f32(f32(src) + f32(0.5)) - sample)
            w[i] = fn(f32(pos * one_over))
            total = f32(total + w[i])
        if abs(total) >= f32(1000.0) * np.finfo(f32).tiny:
            inv = f32(1.0) / total
            w = (w * inv).astype(f32)
        out.append((start, w))
    return out


def resize_antialias(img_f32: np.ndarray, out_h: int, out_w: int, kernel: str) -> np.ndarray:
    """[H, W, C] float32 -> [out_h, out_w, C] float32: tf.image.resize(method, antialias=True)"""
    H, W, C = img_f32.shape
    rows, cols = compute_spans(H, out_h, kernel), compute_spans(W, out_w, kernel)
    inter = np.zeros((out_h, W, C), f32)
    for y, (s, w) in enumerate(rows):
        acc = np.zeros((W, C), f32)
        for i in range(len(w)):
            acc = (acc + (w[i] * img_f32[s + i]).astype(f32)).astype(f32)
        inter[y] = acc
    out = np.zeros((out_h, out_w, C), f32)
    for x, (s, w) in enumerate(cols):
        acc = np.zeros((out_h, C), f32)
        for i in range(len(w)):
            acc = (acc + (w[i] * inter[:, s + i]).astype(f32)).astype(f32)
        out[:, x] = acc
    return out


def to_float01(img_u8: np.ndarray) -> np.ndarray:
    """tf.image.convert_image_dtype(uint8 -> float32): cast, then multiply by float32(1 / 255)"""
    return (img_u8.astype(f32) * f32(1.0 / 255.0)).astype(f32)


def inception_crop_resize(img_u8: np.ndarray, box, size: int) -> np.ndarray:
    """decode_jpeg_and_inception_crop after the decoder, with the sampled box (y, x, h, w) given: float [size, size, 3]"""
    y, x, h, w = box
    return resize_antialias(to_float01(img_u8)[y:y + h, x:x + w], size, size, "triangle")


def resize_small_size(h: int, w: int, size: int):
    """the (new_h, new_w) of resize_small (ops_image.py:104-111): float32 arithmetic, tf.round = round half to even"""
    scale = f32(size) / max(f32(1.0), f32(min(h, w)))
    return int(np.rint(f32(f32(h) * scale))), int(np.rint(f32(f32(w) * scale)))


def crop_or_pad(img: np.ndarray, size: int) -> np.ndarray:
    """tf.image.resize_with_crop_or_pad(image, size, size): central crop of what is larger, zero padding around what is smaller"""
    H, W, C = img.shape
    out = np.zeros((size, size, C), img.dtype)
    cy, cx = max((H - size) // 2, 0), max((W - size) // 2, 0)
    py, px = max((size - H) // 2, 0), max((size - W) // 2, 0)
    hh, ww = min(H, size), min(W, size)
    out[py:py + hh, px:px + ww] = img[cy:cy + hh, cx:cx + ww]
    return out


def eval_resize_crop(img_u8: np.ndarray, resize: int, crop: int, method: str = "keyscubic") -> np.ndarray:
    """decode | resize_small(resize) | central_crop(crop) after the decoder: float [crop, crop, 3]"""
    nh, nw = resize_small_size(img_u8.shape[0], img_u8.shape[1], resize)
    return crop_or_pad(resize_antialias(to_float01(img_u8), nh, nw, method), crop)


def generate_random_crop(width: int, height: int, min_area_frac: float, max_area_frac: float, aspect_ratio: float, uniform):
    """GenerateRandomCrop of sample_distorted_bounding_box_op.cc: (ok, (x0, y0, x1, y1)); `uniform(n)` draws an integer in [0, n)"""
    if max_area_frac <= 0.0 or aspect_ratio <= 0.0:
        return False, None
    min_area = min_area_frac * width * height
    max_area = max_area_frac * width * height
    lrint = lambda v: int(np.rint(v))
    h = lrint(math.sqrt(min_area / aspect_ratio))
    max_h = lrint(math.sqrt(max_area / aspect_ratio))
    if lrint(max_h * aspect_ratio) > width:
        max_h = int((width + 0.5 - 0.0000001) / aspect_ratio)
    if max_h > height:
        max_h = height
    if h >= max_h:
        h = max_h
    if h < max_h:
        h += uniform(max_h - h + 1)
    w = lrint(h * aspect_ratio)
    area = float(w * h)
    if area < min_area:
        h += 1
        w = lrint(h * aspect_ratio)
        area = float(w * h)
    if area > max_area:
        h -= 1
        w = lrint(h * aspect_ratio)
        area = float(w * h)
    if area < min_area or area > max_area or w > width or h > height or w <= 0 or h <= 0:
        return False, None
    y = uniform(height - h) if h < height else 0        # the op draws random->Uniform(original - size): the origin is in [0, H - h)
    x = uniform(width - w) if w < width else 0
    return True, (x, y, x + w, y + h)


def sample_inception_box(h: int, w: int, rng: np.random.Generator, area=(0.08, 1.0), ratio=(0.75, 1.3333333), max_attempts: int = 10):
    """tf.image.sample_distorted_bounding_box(shape, no boxes, min_object_covered=0, use_image_if_no_bounding_boxes=True): up to
    max_attempts draws of an aspect ratio and a crop; the whole image if none fits.  Returns (y, x, h, w)."""
    for _ in range(max_attempts):
        ar = float(rng.uniform(ratio[0], ratio[1]))
        ok, rect = generate_random_crop(w, h, area[0], area[1], ar, lambda n: int(rng.integers(0, n)))
        if ok:
            x0, y0, x1, y1 = rect
            return y0, x0, y1 - y0, x1 - x0
    return 0, 0, h, w
