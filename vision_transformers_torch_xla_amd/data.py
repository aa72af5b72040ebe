"""Device side of the input pipeline.

The reference decodes, crops and resizes on the host with tf.data, then converts every image to float32, transposes it to
CHW and normalises it in Python before the batch is collated and sent to the device as fp32 (datasets.py:254-314,
big_vision/pp/ops_image.py, ops_general.py).  Here the host hands over what the decoder produced - uint8 (or float32) NHWC
batches - through pinned staging buffers on a copy stream, and one HIP kernel (``vitk_preprocess_images``) does flip_lr,
value_range, the layout change and the normalisation on the GPU: 4x fewer bytes cross PCIe for uint8 sources and the host
loop touches no pixel.  JPEG decoding, the inception crop and the resize stay host work (they are not on the measured path).
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


class DeviceLoader:
    """Wraps an iterable of host batches ``(images, labels)`` - images uint8 or float32 ``[B, H, W, 3]`` (numpy or CPU torch),
    labels integer ``[B]`` - and yields what ``train_one_epoch`` / ``evaluate`` consume: ``(samples fp32 [B,3,h,w], targets
    int64 [B])`` on the device, normalised as BigVisionImageNetDataset does (datasets.py:249-252).

    Two pinned staging slots and a copy stream keep one batch in flight while the previous one trains.  ``flip_lr`` draws one
    fair coin per sample (ops_image.py:69-82) from a generator seeded with ``seed + epoch`` (datasets.py:258-261).
    """

    def __init__(self, batches, device, flip_lr=False, value_range=(0.0, 1.0), normalize="imagenet", crop=None, seed=None, length=None):
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
        self._epoch += 1
        self._gen = torch.Generator()
        self._gen.manual_seed((self.seed if self.seed is not None else 0) + self._epoch)
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
            samples = ops.preprocess_images(img_dev, flip=flip_dev, window=window, value_range=self.value_range, mean=mean, std=std)
            yield samples, lab_dev
