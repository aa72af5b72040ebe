#!/usr/bin/env python3
"""LayerNorm launch times at the training shape, with and without the sample maps of stochastic depth.
usage: bench_ln.py [reps] [B] [tokens] [D] [keep]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from vision_transformers_torch_xla_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = int(sys.argv[3]) if len(sys.argv) > 3 else 197
D = int(sys.argv[4]) if len(sys.argv) > 4 else 768
keep = float(sys.argv[5]) if len(sys.argv) > 5 else 0.95
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M = B * T
x = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
dy = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
dres = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
gamma = torch.randn(D, generator=g, device=dev)
beta = torch.randn(D, generator=g, device=dev)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
kept_mask = torch.rand(B, generator=torch.Generator().manual_seed(1)) < keep
kept = torch.nonzero(kept_mask).flatten().to(torch.int32)
inv = torch.full((B,), -1, dtype=torch.int32)
inv[kept.long()] = torch.arange(kept.numel(), dtype=torch.int32)
kept_d, inv_d = kept.to(dev), inv.to(dev)
nk = kept.numel()
scale = (kept_mask.float() / keep).to(dev)
_, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
_, mean_c, rstd_c = ops.layernorm_fwd(x, gamma, beta, smap=kept_d, rows_per_sample=T)
dy_c = dy[: nk * T].contiguous()


def timed(name, f, nbytes):
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
    print(f"{name:34s} {us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s", flush=True)


row = D * 2
timed("fwd", lambda: ops.layernorm_fwd(x, gamma, beta), M * row * 2)
timed(f"fwd gather ({nk}/{B} kept)", lambda: ops.layernorm_fwd(x, gamma, beta, smap=kept_d, rows_per_sample=T), nk * T * row * 2)
timed("bwd res", lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres), M * row * 4)
timed("bwd res+scaled", lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, rowscale=scale, rows_per_group=T), M * row * 5)
timed("bwd res+scaled, in/out maps",
      lambda: ops.layernorm_bwd(dy_c, x, mean_c, rstd_c, gamma, dg, db, dres=dres, rowscale=scale, rows_per_group=T, in_inv=inv_d,
                                out_inv=inv_d, out_rows=nk * T), (M * 3 + nk * T * 2) * row)
