#!/usr/bin/env python3
"""How long the HOST needs to enqueue one training step against how long the GPU needs to run it (one line per workload).

host_ms: wall time of the Python loop body (forward, loss, backward, optimizer step) with no synchronisation inside the
loop, averaged over the steps the host can run ahead; step_ms: the same loop timed to the final synchronize.  Where
host_ms is close to step_ms the workload is bound by launches, not by kernels.

    python tools/host_probe.py [--steps K]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

CASES = [("vit_tiny_patch16_224", 64), ("vit_tiny_patch16_224", 512), ("vit_base_patch16_224", 64), ("vit_base_patch16_224", 512)]


def run(name, batch, steps, drop_path):
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = create_model(name, pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=drop_path).to(dev)
    crit = LabelSmoothingCrossEntropy(0.1)

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(OptArgs, model)
    x = torch.randn(batch, 3, 224, 224, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        step()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(json.dumps({"model": name, "batch": batch, "drop_path": drop_path, "host_ms": round(host / steps * 1e3, 2),
                      "step_ms": round(total / steps * 1e3, 2), "images_per_s": round(batch * steps / total, 1)}), flush=True)
    del model, opt, x, y
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    for name, batch in CASES:
        for dp in (0.0, 0.1):
            run(name, batch, a.steps, dp)
