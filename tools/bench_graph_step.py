#!/usr/bin/env python3
"""engine.train_one_epoch on launch-bound workloads with and without VITK_GRAPH_STEP=1 (forward + loss + backward replayed from a
HIP graph), one process, settings interleaved; both with stochastic depth computed for every sample, the arithmetic the replayable
step uses.

    python tools/bench_graph_step.py [steps]
"""
import contextlib
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["VITK_DROPPATH_COMPACT"] = "0"
import torch

from vision_transformers_torch_xla_amd import engine, optim_factory
from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
from vision_transformers_torch_xla_amd.vit import create_model

CASES = [("vit_tiny_patch16_224", 64), ("my_vit_mini", 64), ("vit_small_patch16_224", 64), ("vit_tiny_patch16_224", 256), ("vit_base_patch16_224", 64)]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    for name, batch in CASES:
        res = {}
        for rnd in range(2):
            for flag in ("0", "1"):
                os.environ["VITK_GRAPH_STEP"] = flag
                torch.manual_seed(0)
                model = create_model(name, pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev)

                class OptArgs:
                    opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
                opt = optim_factory.create_optimizer(OptArgs, model)
                crit = LabelSmoothingCrossEntropy(0.1)
                x = torch.randn(batch, 3, 224, 224, device=dev)
                y = torch.randint(0, 1000, (batch,), device=dev)
                with contextlib.redirect_stdout(sys.stderr):
                    engine.train_one_epoch(model, crit, [(x, y)] * 6, opt, dev, 0, None, 0, None, None, update_freq=1, log_freq=10)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    st = engine.train_one_epoch(model, crit, [(x, y)] * steps, opt, dev, 0, None, 0, None, None, update_freq=1, log_freq=10)
                    torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                res[flag] = min(res.get(flag, 1e9), ms)
                loss = st["loss"]
        print(json.dumps({"model": name, "batch": batch, "eager_ms": round(res["0"], 2), "graphed_ms": round(res["1"], 2),
                          "eager_images_per_s": round(batch / res["0"] * 1e3), "graphed_images_per_s": round(batch / res["1"] * 1e3),
                          "loss": loss}), flush=True)


if __name__ == "__main__":
    main()
