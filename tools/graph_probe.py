#!/usr/bin/env python3
"""Feasibility probe: forward + loss + backward of a launch-bound workload captured in a HIP graph (torch.cuda.graph) and replayed,
against the eager step: same losses / gradients bit for bit, host and wall time per step.

    python tools/graph_probe.py [model] [batch] [steps]
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["VITK_DEFER_WT"] = "0"
import torch

from vision_transformers_torch_xla_amd import optim_factory
from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
from vision_transformers_torch_xla_amd.vit import create_model


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny_patch16_224"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    dev = torch.device("cuda", 0)

    def build():
        torch.manual_seed(0)
        m = create_model(name, pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.0).to(dev)

        class OptArgs:
            opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
        return m, optim_factory.create_optimizer(OptArgs, m)

    crit = LabelSmoothingCrossEntropy(0.1)
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(batch, 3, 224, 224, generator=g, device=dev) for _ in range(4)]
    ys = [torch.randint(0, 1000, (batch,), generator=g, device=dev) for _ in range(4)]

    # eager
    model, opt = build()
    losses_e = []
    for i in range(steps + 3):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = crit(model(xs[i % 4]), ys[i % 4])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses_e.append(loss.detach().clone())
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / steps * 1e3
    p_e = model._flat["p"].clone()

    # graphed: forward + loss + backward in the graph, optimizer step eager
    model, opt = build()
    sx, sy = xs[0].clone(), ys[0].clone()
    losses_g = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on a side stream, as torch's capture recipe asks
        for i in range(3):
            sx.copy_(xs[i % 4]); sy.copy_(ys[i % 4])
            loss = crit(model(sx), sy)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses_g.append(loss.detach().clone())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad()
    with torch.cuda.graph(graph):
        sloss = crit(model(sx), sy)
        sloss.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(3, steps + 3):
        h0 = time.perf_counter()
        sx.copy_(xs[i % 4]); sy.copy_(ys[i % 4])
        graph.replay()
        opt.step()
        losses_g.append(sloss.detach().clone())
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) / steps * 1e3
    same_loss = all(torch.equal(a, b) for a, b in zip(losses_e, losses_g))
    same_p = torch.equal(p_e, model._flat["p"])
    print(f"{name} batch {batch}: eager {eager_ms:.2f} ms/step, graphed {graph_ms:.2f} ms/step (host {host / steps * 1e3:.2f}); "
          f"losses identical: {same_loss}, weights identical after {steps + 3} steps: {same_p}; last loss {losses_g[-1].item():.5f}")


if __name__ == "__main__":
    main()
