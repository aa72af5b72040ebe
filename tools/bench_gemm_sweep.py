#!/usr/bin/env python3
"""gemm_nt over N and epilogue at fixed M, K: separates what an epilogue costs from what the width of the weight panel costs.
usage: python tools/bench_gemm_sweep.py [K] [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import _lib, ops
from bench_gemm import timeit

M = 512 * 197


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    a = rnd(M, K)
    for n in ((3072,) if len(sys.argv) > 3 else (768, 1536, 2304, 3072)):
        b = rnd(n, K)
        bias = torch.randn(n, device=dev)
        out = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
        out2 = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
        res = rnd(M, n)
        for name, epi, kw in (("plain", _lib.EPI_PLAIN, {}), ("bias", _lib.EPI_BIAS, {}), ("bias+res", _lib.EPI_BIAS_RES, dict(res=res)),
                              ("bias+gelu", _lib.EPI_BIAS_GELU, dict(out2=out2)), ("dgelu", _lib.EPI_DGELU, dict(aux=res)),
                              ("bias+gelu+dg", _lib.EPI_BIAS_GELU_DG, dict(out2=out2)), ("mul", _lib.EPI_MUL, dict(aux=res))):
            ms = timeit(lambda: ops.gemm_nt(a, b, epilogue=epi, bias=bias, out=out, **kw), reps)
            tiles = ((M + 255) // 256) * ((n + 255) // 256)
            rounds = -(-tiles // 256)
            print(f"N={n:5d} K={K:5d} {name:10s} {ms * 1e3:7.1f} us  {2.0 * M * n * K / ms / 1e9:7.1f} TFLOP/s  {ms * 1e3 / rounds:6.2f} us/tile-round ({rounds} rounds)", flush=True)


if __name__ == "__main__":
    main()
