#!/usr/bin/env python3
"""The four-wave deferred-epilogue NT kernel (gemm_ntq_kernel) against the eight-wave one (gemm_ntp_kernel, the default; VITK_GEMM_NTQ=1 selects the four-wave kernel) on the
workload's shapes, interleaved rounds in one process: bit equality of the results, then microseconds and TFLOP/s of both.
usage: python tools/ab_ntq.py [rounds] [reps]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import _lib, ops


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    M = int(os.environ.get("AB_M", 486 * 197))          # the kept samples of a stochastic-depth step, about 95 % of 512
    shapes = [("qkv fwd", 2304, 768, _lib.EPI_BIAS), ("proj fwd", 768, 768, _lib.EPI_BIAS_RES), ("fc2 fwd", 768, 3072, _lib.EPI_BIAS_RES),
              ("dfc2", 3072, 768, _lib.EPI_MUL), ("dfc1", 768, 3072, _lib.EPI_PLAIN), ("dproj", 768, 768, _lib.EPI_PLAIN),
              ("dqkv", 768, 2304, _lib.EPI_PLAIN)]
    tot = {"0": 0.0, "1": 0.0}
    for name, N, K, epi in shapes:
        a, b = rnd(M, K), rnd(N, K) * 0.05
        bias = torch.randn(N, device=dev)
        side = rnd(M, N)
        kw = {}
        if epi == _lib.EPI_BIAS_RES:
            kw = dict(res=side, bias=bias)
        elif epi == _lib.EPI_MUL:
            kw = dict(aux=side)
        elif epi == _lib.EPI_BIAS:
            kw = dict(bias=bias)
        outs = {}
        best = {"0": 1e9, "1": 1e9}
        for r in range(rounds):
            for mode in ("0", "1"):
                os.environ["VITK_GEMM_NTQ"] = mode
                out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
                ms = timeit(lambda: ops.gemm_nt(a, b, epilogue=epi, out=out, **kw), reps)
                best[mode] = min(best[mode], ms)
                outs[mode] = out
        same = torch.equal(outs["0"], outs["1"])
        fl = 2.0 * M * N * K
        for mode in ("0", "1"):
            tot[mode] += best[mode]
        print(f"{name:9s} N={N:5d} K={K:5d}: eight-wave {best['0'] * 1e3:7.1f} us {fl / best['0'] / 1e9:7.1f} TF | four-wave {best['1'] * 1e3:7.1f} us "
              f"{fl / best['1'] / 1e9:7.1f} TF | {best['0'] / best['1']:.3f}x | bit-identical: {same}", flush=True)
    print(f"sum over the shapes: eight-wave {tot['0'] * 1e3:.1f} us, four-wave {tot['1'] * 1e3:.1f} us ({tot['0'] / tot['1']:.3f}x)")


if __name__ == "__main__":
    main()
