#!/usr/bin/env python3
"""Per-step wall time next to the caching allocator's segment count, for several models trained one after another in one
process: does a slow step coincide with a fresh hipMalloc?  usage: alloc_probe.py steps model:batch [model:batch ...]"""
import argparse
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from vision_transformers_torch_xla_amd import losses, optim_factory
from vision_transformers_torch_xla_amd.vit import create_model

dev = torch.device("cuda:0")
n = int(sys.argv[1])
for spec in sys.argv[2:]:
    name, batch = spec.split(":")
    batch = int(batch)
    torch.manual_seed(0)
    model = create_model(name, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev)
    model.train()
    opt = optim_factory.create_optimizer(argparse.Namespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None), model)
    crit = losses.LabelSmoothingCrossEntropy(0.1)
    size = model.patch_embed.img_size[0]
    x = torch.randn(batch, 3, size, size, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)
    prev = None
    print(f"== {name} batch {batch}", flush=True)
    for i in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) * 1e3
        st = torch.cuda.memory_stats()
        cur = (st["segment.all.allocated"], st["reserved_bytes.all.current"] >> 20)
        print(f"step {i:3d} {dt:7.2f} ms  segments {cur[0]} reserved {cur[1]} MiB{'  <-- new segment' if prev is not None and cur != prev else ''}", flush=True)
        prev = cur
    del model, opt, x, y, loss
    torch.cuda.empty_cache()
