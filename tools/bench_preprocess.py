#!/usr/bin/env python3
"""Launch time of the input pipeline tail at the training batch.  usage: bench_preprocess.py [reps] [B] [S]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from vision_transformers_torch_xla_amd import data, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
S = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = torch.device("cuda:0")
flip = (torch.rand(B, device=dev) < 0.5).to(torch.uint8)
for kind in ("u8", "f32"):
    src = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device=dev) if kind == "u8" else torch.rand(B, S, S, 3, device=dev)
    out = torch.empty(B, 3, S, S, device=dev)
    f = lambda: ops.preprocess_images(src, flip=flip, out=out, mean=data.IMAGENET_DEFAULT_MEAN, std=data.IMAGENET_DEFAULT_STD)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    nbytes = src.numel() * src.element_size() + out.numel() * 4
    print(f"{kind}: {us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s  ({B}x{S}x{S})", flush=True)


def bench_resample():
    """crop + antialiased resize + flip + normalise of 512 ImageNet-sized images (random sizes around 500 x 375) to 224 x 224"""
    import numpy as np
    from vision_transformers_torch_xla_amd import data
    rng = np.random.default_rng(0)
    dev = torch.device("cuda:0")
    images = [rng.integers(0, 256, (int(rng.integers(300, 520)), int(rng.integers(300, 520)), 3), dtype=np.uint8) for _ in range(512)]
    flat, offs = data.pack_images(images)
    src = torch.from_numpy(flat).to(dev)
    for name, desc, kernel in (("train: inception crop -> 224 (triangle)", data.train_descriptors(images, offs, 224, rng), "triangle"),
                               ("eval: resize_small(256) | central_crop(224) (keyscubic)", data.eval_descriptors(images, offs, 256, 224), "keyscubic")):
        dd = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev)
        out = torch.empty((512, 3, 224, 224), device=dev)
        f = lambda: ops.resample_images(src, desc, 224, kernel=kernel, out=out, mean=data.IMAGENET_DEFAULT_MEAN, std=data.IMAGENET_DEFAULT_STD, desc_dev=dd)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"resample {name}: {ms * 1e3:8.1f} us / 512 images = {512 / ms * 1e3:,.0f} images/s ({flat.size / 1e6:.0f} MB of pixels in)", flush=True)


if __name__ == "__main__":
    bench_resample()
