#!/usr/bin/env python3
"""Every epilogue of the persistent NT GEMM on ONE shape (default M = 512 * 197, N = K = 768: 5 tile rounds, 12 K-steps per tile), so
that what an epilogue costs per tile can be read off against the plain one.  usage: python tools/bench_epilogues.py [N] [K] [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import _lib, ops
from tools.bench_gemm import timeit

M = 512 * 197


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    a, b = rnd(M, k), rnd(n, k)
    bias = torch.randn(n, device=dev)
    res, aux = rnd(M, n), rnd(M, n)
    out, out2 = torch.empty((M, n), dtype=torch.bfloat16, device=dev), torch.empty((M, n), dtype=torch.bfloat16, device=dev)
    tiles = -(-M // 256) * -(-n // 256)
    rounds = -(-tiles // 256)
    base = None
    for _ in range(20):          # bring the clocks to their state under load before the first timed launch
        ops.gemm_nt(a, b, epilogue=_lib.EPI_PLAIN, out=out)
    for name, epi, kw in (("plain", _lib.EPI_PLAIN, {}), ("bias", _lib.EPI_BIAS, dict(bias=bias)),
                          ("bias+res", _lib.EPI_BIAS_RES, dict(bias=bias, res=res)), ("mul", _lib.EPI_MUL, dict(aux=aux)),
                          ("bias+gelu+dg", _lib.EPI_BIAS_GELU_DG, dict(bias=bias, out2=out2)), ("plain again", _lib.EPI_PLAIN, {})):
        ms = timeit(lambda: ops.gemm_nt(a, b, epilogue=epi, out=out, **kw), reps)
        base = base or ms
        print(f"N={n} K={k} {name:13s} {ms * 1e3:8.1f} us  {2.0 * M * n * k / ms / 1e9:7.1f} TFLOP/s   {ms * 1e3 / rounds:6.2f} us per tile round"
              f"   {(ms - base) * 1e3 / rounds:+6.2f} vs plain", flush=True)


if __name__ == "__main__":
    main()
