#!/usr/bin/env python3
"""How much does a collective-like kernel that holds a few compute units for a while (tools/probe/occupy.hip) slow the
training step down?  The persistent GEMMs assign tiles to workgroups statically, so a workgroup whose compute unit is held
starts late and finishes the launch late.  Build the stand-in first:
    hipcc -O2 -fPIC -shared --offload-arch=gfx950 -o tools/probe/liboccupy.so tools/probe/occupy.hip
usage: occupy_probe.py [steps]"""
import argparse
import ctypes
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from vision_transformers_torch_xla_amd import losses, optim_factory
from vision_transformers_torch_xla_amd.vit import create_model

occ = ctypes.CDLL(str(ROOT / "tools" / "probe" / "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
model = create_model("vit_base_patch16_224", num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev)
model.train()
opt = optim_factory.create_optimizer(argparse.Namespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None), model)
crit = losses.LabelSmoothingCrossEntropy(0.1)
x = torch.randn(512, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (512,), device=dev)
comm = torch.cuda.Stream()


def make_hook(wgs, us, every):
    state = {"n": 0}

    def hook(upto, final, after=None):
        state["n"] += 1
        if wgs and (final or state["n"] % every == 0):
            ev = torch.cuda.Event()
            ev.record()
            comm.wait_event(ev)
            occ.occupy(wgs, int(us * 100), comm.cuda_stream)
        if final:
            torch.cuda.current_stream().wait_stream(comm)
    return hook


def run(label, hook):
    model._grad_ready_hook = hook
    for phase, n in (("warm", 4), ("timed", steps)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            crit(model(x), y).backward()
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n * 1e3
    print(f"{label:44s} {dt:7.2f} ms/step", flush=True)


run("no hook (side stream on)", None)
run("hook, no occupier (one stream)", make_hook(0, 0, 1))
for wgs, us, every in ((16, 300, 1), (32, 300, 1), (32, 1000, 1), (8, 300, 1), (8, 600, 2), (16, 600, 2)):
    run(f"occupier {wgs} workgroups x {us} us, {13 // every + (every > 1)} per step", make_hook(wgs, us, every))
