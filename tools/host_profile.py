#!/usr/bin/env python3
"""cProfile of the host side of a launch-bound training step (ViT-Tiny/16, 64 images): where the Python time per step goes.

    python tools/host_profile.py [steps]
"""
import cProfile
import os
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

from vision_transformers_torch_xla_amd import optim_factory
from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
from vision_transformers_torch_xla_amd.vit import create_model


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = create_model("vit_tiny_patch16_224", pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev)
    crit = LabelSmoothingCrossEntropy(0.1)

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(OptArgs, model)
    x = torch.randn(64, 3, 224, 224, device=dev)
    y = torch.randint(0, 1000, (64,), device=dev)

    def step():
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumtime").print_stats(22)


if __name__ == "__main__":
    main()
