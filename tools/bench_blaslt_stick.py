#!/usr/bin/env python3
"""Measuring stick only (never on the product path): the vendor GEMM library behind torch.matmul on the same random bf16 operands,
for the square sizes and the workload's shapes, beside this package's gemm_nt.  usage: python tools/bench_blaslt_stick.py [reps]"""
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
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (100864, 2304, 768), (100864, 768, 768), (100864, 3072, 768), (100864, 768, 3072),
              (100864, 768, 2304)]
    for (m, n, k) in shapes:
        a = torch.randn(m, k, generator=g, device=dev).to(torch.bfloat16)
        b = torch.randn(n, k, generator=g, device=dev).to(torch.bfloat16)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        bt = b.t()
        for _ in range(2):
            ms_lib = timeit(lambda: torch.matmul(a, bt, out=out), reps)
            ms_own = timeit(lambda: ops.gemm_nt(a, b, out=out), reps)
        fl = 2.0 * m * n * k / 1e9
        print(f"M={m:6d} N={n:5d} K={k:5d}   vendor library {ms_lib * 1e3:8.1f} us {fl / ms_lib:7.1f} TFLOP/s   gemm_nt {ms_own * 1e3:8.1f} us {fl / ms_own:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
