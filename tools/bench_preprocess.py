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
