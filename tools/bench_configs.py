#!/usr/bin/env python3
"""Throughput of the other single-GPU workloads (ViT-L/16 224, ViT-B/16 384, DeiT-B/16 distilled + knowledge distillation)
through the same engine.train_one_epoch loop bench.py uses.  One line per workload on stdout.

    python tools/bench_configs.py [--only NAME] [--steps K] [--warmup W]
"""
import argparse
import contextlib
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

from bench import SyntheticLoader

# name -> (student, per-GPU batch, image size, train GFLOP / image (6 x forward MACs), teacher or None)
WORKLOADS = {
    "vit_b_224": ("vit_base_patch16_224", 512, 224, 105.38, None),
    "vit_l_224": ("vit_large_patch16_224", 256, 224, 369.33, None),
    "vit_b_384": ("vit_base_patch16_384", 128, 384, 332.91, None),
    # the reference's CPU-runnable case and its small sizes (models/my_vit.py:84-165; head dims 48 / 64 / 72)
    "vit_tiny_224": ("vit_tiny_patch16_224", 512, 224, 7.52, None),
    "my_vit_mini": ("my_vit_mini", 512, 224, 4.47, None),
    "my_vit_xs": ("my_vit_xs", 512, 224, 15.99, None),
    "deit_b_kd": ("deit_base_distilled_patch16_224", 512, 224, 105.94 + 35.13, "vit_base_patch16_224"),
    # the reference's default recipe (main.py:219-229, run_train.sh:64-65): Mixup 0.8 / CutMix 1.0 on the device, soft-target CE, EMA
    "vit_b_224_recipe": ("vit_base_patch16_224", 512, 224, 105.38, "recipe"),
    # engine.evaluate: forward + CE + top-1 / top-5, batch 1.5 x 512 (main.py:603); 2 x forward MACs per image
    "vit_b_224_eval": ("vit_base_patch16_224", 768, 224, 35.13, "eval"),
}


def run(name, steps, warmup):
    from vision_transformers_torch_xla_amd import engine, optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    from vision_transformers_torch_xla_amd.vit import create_model
    student, batch, size, gflop, teacher = WORKLOADS[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pool = "token" if student.startswith("deit") else "avg"
    core = create_model(student, pretrained=False, num_classes=1000, global_pool=pool, drop_path_rate=0.1).to(dev)
    model, crit = core, LabelSmoothingCrossEntropy(0.1)
    mixup_fn = ema = None
    if teacher == "recipe":
        from vision_transformers_torch_xla_amd.losses import SoftTargetCrossEntropy
        from vision_transformers_torch_xla_amd.mixup import Mixup
        from vision_transformers_torch_xla_amd.utils import ModelEma
        mixup_fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, mode="batch", label_smoothing=0.1, num_classes=1000)
        crit = SoftTargetCrossEntropy()
        ema = ModelEma(core, decay=0.9999)
        teacher = None
    elif teacher == "eval":
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(batch, 3, size, size, generator=g, device=dev)
        y = torch.randint(0, 1000, (batch,), generator=g, device=dev)
        with contextlib.redirect_stdout(sys.stderr):
            engine.evaluate([(x, y)] * warmup, core, dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stats = engine.evaluate([(x, y)] * steps, core, dev)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ips = batch * steps / dt
        print(json.dumps({"workload": name, "model": student, "teacher": None, "per_gpu_batch": batch, "images_per_s": round(ips, 1),
                          "ms_per_step": round(dt / steps * 1e3, 2), "tflops": round(ips * gflop / 1e3, 1), "loss": stats.get("loss"),
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
        return
    if teacher:
        t = create_model(teacher, pretrained=False, num_classes=1000, global_pool="avg").to(dev).eval()
        for p in t.parameters():
            p.requires_grad_(False)
        model = StudentWithDistillation(core, t)
        crit = DistillationLoss(crit, alpha=0.7, temperature=4.0)

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 4e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(OptArgs, core)
    total = steps + warmup
    lr = utils.cosine_scheduler(4e-3, 1e-6, 1, total, warmup_steps=max(1, warmup), warmup_epochs=1)
    wd = utils.cosine_scheduler(0.05, 0.05, 1, total)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(batch, 3, size, size, generator=g, device=dev)
    y = torch.randint(0, 1000, (batch,), generator=g, device=dev)

    def go(n, start):
        with contextlib.redirect_stdout(sys.stderr):
            return engine.train_one_epoch(model, crit, SyntheticLoader(x, y, n), opt, dev, 0, None, 0, ema, mixup_fn, start_steps=start,
                                          lr_schedule_values=lr, wd_schedule_values=wd, num_training_steps_per_epoch=n,
                                          update_freq=1, log_freq=10)
    go(warmup, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = go(steps, warmup)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ips = batch * steps / dt
    print(json.dumps({"workload": name, "model": student, "teacher": teacher, "per_gpu_batch": batch, "images_per_s": round(ips, 1),
                      "ms_per_step": round(dt / steps * 1e3, 2), "tflops": round(ips * gflop / 1e3, 1), "loss": stats.get("loss"),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
    del model, core, opt, x, y
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    for n in ([a.only] if a.only else list(WORKLOADS)):
        run(n, a.steps, a.warmup)
