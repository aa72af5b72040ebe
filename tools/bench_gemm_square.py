#!/usr/bin/env python3
"""gemm_nt (plain epilogue) on square problems with random operands: the sizes GEMM kernels are usually quoted on, to place
this kernel's main loop beside published MI355X figures.  usage: python tools/bench_gemm_square.py [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import ops
from bench_gemm import timeit


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for fill in ("uniform", "normal", "zeros"):
        for n in (4096, 8192):
            if fill == "zeros":
                a = torch.zeros(n, n, dtype=torch.bfloat16, device=dev); b = torch.zeros(n, n, dtype=torch.bfloat16, device=dev)
            elif fill == "uniform":
                a = (torch.rand(n, n, generator=g, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, n, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
            else:
                a = torch.randn(n, n, generator=g, device=dev).to(torch.bfloat16); b = torch.randn(n, n, generator=g, device=dev).to(torch.bfloat16)
            out = torch.empty((n, n), dtype=torch.bfloat16, device=dev)
            for _ in range(3):
                ms = timeit(lambda: ops.gemm_nt(a, b, out=out), reps)
            print(f"{fill:8s} {n}^3  {ms * 1e3:8.1f} us  {2.0 * n ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # the same on a workload shape whose activation panel streams from HBM
    for (m, n, k) in ((100864, 768, 3072), (100864, 3072, 768)):
        a = torch.randn(m, k, generator=g, device=dev).to(torch.bfloat16); b = torch.randn(n, k, generator=g, device=dev).to(torch.bfloat16)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            ms = timeit(lambda: ops.gemm_nt(a, b, out=out), reps)
        print(f"normal   M={m} N={n} K={k}  {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
