#!/usr/bin/env python3
"""Per-step wall time next to the caching allocator's segment count: does a slow step coincide with a fresh hipMalloc?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from vision_transformers_torch_xla_amd import losses, optim_factory
from vision_transformers_torch_xla_amd.vit import create_model

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = create_model("vit_base_patch16_224", num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev)
model.train()
import argparse
args = argparse.Namespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None, momentum=0.9)
opt = optim_factory.create_optimizer(args, model)
crit = losses.LabelSmoothingCrossEntropy(0.1)
x = torch.randn(512, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (512,), device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prev = None
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
    print(f"step {i:3d} {dt:7.2f} ms  segments allocated {cur[0]} reserved {cur[1]} MiB {'  <-- new segment' if prev is not None and cur != prev else ''}", flush=True)
    prev = cur
