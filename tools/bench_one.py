#!/usr/bin/env python3
"""One GEMM shape, a few launches: the target of rocprofv3 --pmc passes.  usage: bench_one.py nt|tn M N K [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from vision_transformers_torch_xla_amd import ops
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
if kind == "nt":
    a, b = rnd(M, K), rnd(N, K)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm_nt(a, b, out=out)
else:
    a, b = rnd(M, N), rnd(M, K)
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    f = lambda: ops.gemm_tn(a, b, out)
for _ in range(reps):
    f()
torch.cuda.synchronize()
print("done")
