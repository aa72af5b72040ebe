"""Device side of the input pipeline.

The reference decodes, crops and resizes on the host with tf.data, then converts every image to float32, transposes it to
CHW and normalises it in Python before the batch is collated and sent to the device as fp32 (datasets.py:254-314,
big_vision/pp/ops_image.py, ops_general.py).  Here the host hands over what the decoder produced - uint8 (or float32) NHWC
batches - through pinned staging buffers on a copy stream, and one HIP kernel (``vitk_preprocess_images``) does flip_lr,
value_range, the layout change and the normalisation on the GPU: 4x fewer bytes cross PCIe for uint8 sources and the host
loop touches no pixel.

The crop / resize stage runs on the GPU too (``ResampleLoader``): the host only decodes (``DecodedImageFolder``, PIL) and draws
the per-image geometry - the inception crop box of decode_jpeg_and_inception_crop (ops_image.py:36-66) for training, the
resize_small / central_crop sizes (:85-134) for evaluation - and one kernel (``vitk_resample_images``) does the antialiased
tf.image.resize arithmetic, the crop, flip_lr, value_range and the normalisation for a whole batch of differently sized images.
TFDS shard reading and TensorFlow's JPEG decoder are not rebuilt: images come from an image-folder tree.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

# timm.data.constants, as used at datasets.py:130-131
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def central_crop_window(h: int, w: int, size: int):
    """(top, left, size, size) of central_crop(size) for images no smaller than the crop (ops_image.py:121-134)"""
    if h < size or w < size:
        raise ValueError(f"central crop {size} needs images of at least that size, got {h}x{w} (padding is host work)")
    return (h - size) // 2, (w - size) // 2, size, size


def sample_inception_box(h: int, w: int, rng: np.random.Generator, area=(0.08, 1.0), ratio=(0.75, 1.3333333), max_attempts: int = 10):
    """The crop box (y, x, h, w) that decode_jpeg_and_inception_crop draws (ops_image.py:50-58): tf.image.sample_distorted_
    bounding_box without boxes - up to max_attempts times an aspect ratio uniform in `ratio`, a height between the ones that
    give the smallest and the largest allowed area at that ratio, a uniform position; the whole image when no attempt fits.
    The draws come from `rng` (numpy), not from TensorFlow's stream."""
    lo_area, hi_area = area[0] * w * h, area[1] * w * h
    for _ in range(max_attempts):
        ar = float(rng.uniform(ratio[0], ratio[1]))
        bh, top_h = int(np.rint(np.sqrt(lo_area / ar))), int(np.rint(np.sqrt(hi_area / ar)))
        if int(np.rint(top_h * ar)) > w:
            top_h = int((w + 0.5 - 1e-7) / ar)
        top_h = min(top_h, h)
        bh = min(bh, top_h)
        if bh < top_h:
            bh += int(rng.integers(0, top_h - bh + 1))
        bw = int(np.rint(bh * ar))
        if bw * bh < lo_area:                      # rounding took the area under / over its bound: one step back in
            bh += 1
            bw = int(np.rint(bh * ar))
        if bw * bh > hi_area:
            bh -= 1
            bw = int(np.rint(bh * ar))
        if not (lo_area <= bw * bh <= hi_area) or bw > w or bh > h or bw <= 0 or bh <= 0:
            continue
        y = int(rng.integers(0, h - bh)) if bh < h else 0          # the origin is uniform over [0, H - h), as TensorFlow's
        x = int(rng.integers(0, w - bw)) if bw < w else 0          # GenerateRandomCrop draws it (Uniform(H - h))
        return y, x, bh, bw
    return 0, 0, h, w


def resize_small_size(h: int, w: int, size: int):
    """(new_h, new_w) of resize_small(size) (ops_image.py:104-111): float32 products, ties to even as tf.round"""
    scale = np.float32(size) / max(np.float32(1.0), np.float32(min(h, w)))
    return int(np.rint(np.float32(h) * scale)), int(np.rint(np.float32(w) * scale))


MAX_SOURCE_WIDTH = 12288        # widest crop box one workgroup of the resample kernel keeps in LDS
MAX_ROW_REDUCTION = 100         # largest source-rows : output-rows ratio whose filter taps fit the kernel's table (cubic: 127)


def fit_for_resample(im, size):
    """An image the resample kernel cannot take whole - wider than MAX_SOURCE_WIDTH, or so tall that one output row of `size`
    would need more filter taps than the kernel holds - is halved with a 2 x 2 box average (rounded to nearest) until it fits,
    instead of aborting the epoch; every other image is returned as it is.  (The reference has no such limit: its host resize
    takes any size.  Such images - above 100 megapixels at size 224 - do not occur in ImageNet.)"""
    while im.shape[1] > MAX_SOURCE_WIDTH or im.shape[0] > MAX_ROW_REDUCTION * size:
        h, w = im.shape[0] // 2 * 2, im.shape[1] // 2 * 2
        q = im[:h, :w].astype(np.uint16)
        im = ((q[0::2, 0::2] + q[1::2, 0::2] + q[0::2, 1::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    return im


def pack_images(images):
    """list of uint8 [H, W, 3] arrays -> (one flat uint8 array, byte offsets); every image starts on a 16-byte boundary"""
    offs, total = [], 0
    for im in images:
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError(f"pack_images: expected uint8 [H, W, 3] images, got {im.dtype} {im.shape}")
        offs.append(total)
        total += (im.size + 15) // 16 * 16
    flat = np.zeros(total, dtype=np.uint8)
    for im, o in zip(images, offs):
        flat[o:o + im.size] = np.ascontiguousarray(im).reshape(-1)
    return flat, offs


def train_descriptors(images, offsets, size, rng, flip_lr=True, boxes=None, flips=None):
    """descriptors of decode_jpeg_and_inception_crop(size) | flip_lr; boxes / flips override the draws (tests)"""
    d = np.zeros(len(images), dtype=ops.RESAMPLE_DESC)
    for b, im in enumerate(images):
        h, w = im.shape[:2]
        y, x, bh, bw = boxes[b] if boxes is not None else sample_inception_box(h, w, rng)
        fl = (flips[b] if flips is not None else (rng.random() < 0.5)) if flip_lr else False
        d[b] = (offsets[b], h, w, y, x, bh, bw, size, size, 0, 0, int(bool(fl)), 0)
    return d


def eval_descriptors(images, offsets, resize, crop):
    """descriptors of resize_small(resize) | central_crop(crop): the whole image to (new_h, new_w), then the central window
    (offsets of tf.image.resize_with_crop_or_pad; negative = that much zero padding)"""
    d = np.zeros(len(images), dtype=ops.RESAMPLE_DESC)
    for b, im in enumerate(images):
        h, w = im.shape[:2]
        nh, nw = resize_small_size(h, w, resize)
        wy = (nh - crop) // 2 if nh >= crop else -((crop - nh) // 2)
        wx = (nw - crop) // 2 if nw >= crop else -((crop - nw) // 2)
        d[b] = (offsets[b], h, w, 0, 0, h, w, nh, nw, wy, wx, 0, 0)
    return d


class DecodedImageFolder:
    """Host stage: an image-folder tree (root/<class name>/<file>) decoded to uint8 RGB by PIL on a thread pool, batched as
    (list of [H, W, 3] arrays, int64 labels).  Classes are the sorted sub-directory names.  With `rank` / `world_size` every
    rank reads its own contiguous share of the (per-epoch shuffled) file list, the role of tfds.even_splits at
    datasets.py:174-176.  drop_last as the reference's DataLoader (main.py:567)."""

    EXTENSIONS = (".jpg", ".jpeg", ".png", ".bmp", ".webp")

    def __init__(self, root, batch_size, shuffle=False, seed=0, workers=8, rank=0, world_size=1, drop_last=True):
        import os
        self.root, self.batch_size, self.shuffle, self.seed, self.workers = root, int(batch_size), shuffle, seed, workers
        self.rank, self.world_size, self.drop_last = rank, world_size, drop_last
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise ValueError(f"DecodedImageFolder: no class directories under {root}")
        self.samples = [(os.path.join(root, c, f), k) for k, c in enumerate(self.classes)
                        for f in sorted(os.listdir(os.path.join(root, c))) if f.lower().endswith(self.EXTENSIONS)]
        self._epoch = 0

    def set_epoch(self, epoch: int):
        self._epoch = int(epoch)

    def _share(self):
        order = np.arange(len(self.samples))
        if self.shuffle:
            order = np.random.default_rng(self.seed + self._epoch).permutation(len(self.samples))
        per = len(order) // self.world_size
        return order[self.rank * per:(self.rank + 1) * per]

    def __len__(self):
        n = len(self.samples) // self.world_size
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    @staticmethod
    def _decode(path):
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        order = self._share()
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            for i in range(0, len(order), self.batch_size):
                idx = order[i:i + self.batch_size]
                if len(idx) < self.batch_size and self.drop_last:
                    break
                images = list(pool.map(self._decode, [self.samples[j][0] for j in idx]))
                yield images, np.array([self.samples[j][1] for j in idx], dtype=np.int64)


class ResampleLoader:
    """Wraps an iterable of host batches (list of decoded uint8 [H, W, 3] images of any sizes, labels) and yields
    (samples fp32 [B, 3, size, size], targets int64 [B]) on the device: the reference's whole per-image pipeline after the
    decoder, on the GPU.
      train=True   decode_jpeg_and_inception_crop(size) | flip_lr | value_range(0, 1)   (main.py:272)
      train=False  decode | resize_small(resize) | central_crop(size) | value_range(0, 1)   (main.py:275)
    followed by the ImageNet normalisation (datasets.py:249-252).  Crop boxes and flips are drawn per epoch from
    numpy's Generator(seed + epoch) (datasets.py:258-261 seeds tf.data the same way).  The packed pixels and the descriptors
    cross PCIe through two pinned slots on a copy stream while the previous batch trains.
    patch=p yields the model's bf16 patch matrix [B * (size/p)^2, 3 * p * p] instead of the fp32 batch (the same values the model
    would form from it; for runs whose images nothing else touches between the loader and the model: evaluation, training
    without Mixup / CutMix, which mix fp32 images)."""

    def __init__(self, batches, device, size=224, train=True, resize=256, flip_lr=True, value_range=(0.0, 1.0), normalize="imagenet",
                 seed=None, length=None, method=None, patch=None):
        if normalize not in ("imagenet", "none"):
            raise ValueError(f"normalize must be 'imagenet' or 'none', got {normalize!r}")
        self.patch = patch
        self.batches, self.device, self.size, self.train, self.resize = batches, torch.device(device), int(size), train, int(resize)
        self.flip_lr, self.value_range, self.normalize, self.seed, self._length = flip_lr, value_range, normalize, seed, length
        self.method = method or ("triangle" if train else "keyscubic")      # BILINEAR for the inception crop, BICUBIC for resize_small
        self._epoch, self._copy_stream, self._slots = 0, None, [None, None]

    def __len__(self):
        return self._length if self._length is not None else len(self.batches)

    def set_epoch(self, epoch: int):
        self._epoch = int(epoch)
        if hasattr(self.batches, "set_epoch"):
            self.batches.set_epoch(epoch)

    def _stage(self, slot, images, labels):
        images = [fit_for_resample(im, self.size) for im in images]
        flat, offs = pack_images(images)
        desc = (train_descriptors(images, offs, self.size, self._rng, self.flip_lr) if self.train
                else eval_descriptors(images, offs, self.resize, self.size))
        labels = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(torch.int64)
        dbytes = desc.view(np.uint8).reshape(-1)
        st = self._slots[slot]
        if st is None or st["pix"].numel() < flat.size or st["lab"].numel() != labels.numel():
            st = {"pix": torch.empty(max(flat.size, 1) * 5 // 4, dtype=torch.uint8).pin_memory(),
                  "desc": torch.empty(dbytes.size, dtype=torch.uint8).pin_memory(), "lab": torch.empty(labels.numel(), dtype=torch.int64).pin_memory(),
                  "free": None}
            self._slots[slot] = st
        if st["free"] is not None:
            st["free"].synchronize()
        st["pix"][:flat.size].copy_(torch.from_numpy(flat))
        st["desc"].copy_(torch.from_numpy(dbytes.copy()))
        st["lab"].copy_(labels)
        with torch.cuda.stream(self._copy_stream):
            pix = st["pix"][:flat.size].to(self.device, non_blocking=True)
            dd = st["desc"].to(self.device, non_blocking=True)
            lab = st["lab"].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        st["free"] = ev
        return pix, dd, lab, desc, ev

    def __iter__(self):
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        # the draws of epoch e come from Generator(seed + e); without set_epoch the counter advances by itself
        self._rng = np.random.default_rng((self.seed if self.seed is not None else 0) + self._epoch)
        self._epoch += 1
        mean, std = (IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD) if self.normalize == "imagenet" else (None, None)
        it = iter(self.batches)
        slot = 0
        try:
            pending = self._stage(slot, *next(it))
        except StopIteration:
            return
        while pending is not None:
            pix, dd, lab, desc, ev = pending
            slot ^= 1
            try:
                pending = self._stage(slot, *next(it))      # decode, pack and copy the next batch while this one is consumed
            except StopIteration:
                pending = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (pix, dd, lab):
                t.record_stream(cur)
            yield ops.resample_images(pix, desc, self.size, kernel=self.method, value_range=self.value_range, mean=mean, std=std, desc_dev=dd,
                                      patch=self.patch), lab


class DeviceLoader:
    """Wraps an iterable of host batches ``(images, labels)`` - images uint8 or float32 ``[B, H, W, 3]`` (numpy or CPU torch),
    labels integer ``[B]`` - and yields what ``train_one_epoch`` / ``evaluate`` consume: ``(samples fp32 [B,3,h,w], targets
    int64 [B])`` on the device, normalised as BigVisionImageNetDataset does (datasets.py:249-252).

    Two pinned staging slots and a copy stream keep one batch in flight while the previous one trains.  ``flip_lr`` draws one
    fair coin per sample (ops_image.py:69-82) from a generator seeded with ``seed + epoch`` (datasets.py:258-261).
    """

    def __init__(self, batches, device, flip_lr=False, value_range=(0.0, 1.0), normalize="imagenet", crop=None, seed=None, length=None, patch=None):
        self.patch = patch              # yield the model's bf16 patch matrix instead of the fp32 batch (see ResampleLoader)
        if normalize not in ("imagenet", "none"):
            raise ValueError(f"normalize must be 'imagenet' or 'none', got {normalize!r}")
        self.batches = batches
        self.device = torch.device(device)
        self.flip_lr = flip_lr
        self.value_range = value_range
        self.normalize = normalize
        self.crop = crop
        self.seed = seed
        self._epoch = 0
        self._length = length
        self._copy_stream = None
        self._slots = [None, None]

    def __len__(self):
        return self._length if self._length is not None else len(self.batches)

    def set_epoch(self, epoch: int):
        self._epoch = int(epoch)

    def _stage(self, slot, images, labels):
        """async copy of one host batch into device buffers on the copy stream; returns (images_dev, labels_dev, flip_dev, event)"""
        images = torch.from_numpy(np.ascontiguousarray(images)) if isinstance(images, np.ndarray) else images.contiguous()
        labels = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(torch.int64)
        B = images.shape[0]
        st = self._slots[slot]
        if st is None or st["img"].shape != images.shape or st["img"].dtype != images.dtype:
            st = {"img": torch.empty(images.shape, dtype=images.dtype).pin_memory(), "lab": torch.empty(B, dtype=torch.int64).pin_memory(),
                  "flip": torch.empty(B, dtype=torch.uint8).pin_memory(), "free": None}
            self._slots[slot] = st
        if st["free"] is not None:
            st["free"].synchronize()            # the copies that last read this slot have finished
        st["img"].copy_(images)
        st["lab"].copy_(labels)
        flip_dev = None
        with torch.cuda.stream(self._copy_stream):
            img_dev = st["img"].to(self.device, non_blocking=True)
            lab_dev = st["lab"].to(self.device, non_blocking=True)
            if self.flip_lr:
                st["flip"].copy_((torch.rand(B, generator=self._gen) < 0.5).to(torch.uint8))
                flip_dev = st["flip"].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        st["free"] = ev
        return img_dev, lab_dev, flip_dev, ev

    def __iter__(self):
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        self._gen = torch.Generator()           # epoch e flips with Generator(seed + e); without set_epoch the counter advances by itself
        self._gen.manual_seed((self.seed if self.seed is not None else 0) + self._epoch)
        self._epoch += 1
        mean, std = (IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD) if self.normalize == "imagenet" else (None, None)
        it = iter(self.batches)
        pending, slot = None, 0
        try:
            pending = self._stage(slot, *next(it))
        except StopIteration:
            return
        while pending is not None:
            img_dev, lab_dev, flip_dev, ev = pending
            slot ^= 1
            try:
                pending = self._stage(slot, *next(it))     # the next batch crosses PCIe while this one is consumed
            except StopIteration:
                pending = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (img_dev, lab_dev, flip_dev):
                if t is not None:
                    t.record_stream(cur)
            window = None
            if self.crop is not None:
                window = central_crop_window(img_dev.shape[1], img_dev.shape[2], self.crop)
            samples = ops.preprocess_images(img_dev, flip=flip_dev, window=window, value_range=self.value_range, mean=mean, std=std,
                                            patch=self.patch)
            yield samples, lab_dev
