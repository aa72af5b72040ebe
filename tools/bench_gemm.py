#!/usr/bin/env python3
"""Micro-benchmark of the libvitk GEMMs on the ViT-B/16 (B=512) shapes: TFLOP/s per shape, random bf16 operands.
usage: python tools/bench_gemm.py [nt|tn|all] [reps] [nodb]     (nodb: weight gradients without the bias gradient)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import _lib, ops

M = 512 * 197
NT = [("qkv fwd", M, 2304, 768, _lib.EPI_BIAS), ("proj fwd", M, 768, 768, _lib.EPI_BIAS_RES), ("fc1 fwd", M, 3072, 768, _lib.EPI_BIAS_GELU_DG),
      ("fc2 fwd", M, 768, 3072, _lib.EPI_BIAS_RES), ("dqkv->dy", M, 768, 2304, _lib.EPI_PLAIN), ("dproj->do", M, 768, 768, _lib.EPI_PLAIN),
      ("dfc2->dh", M, 3072, 768, _lib.EPI_MUL), ("dfc1->dy", M, 768, 3072, _lib.EPI_PLAIN)]
TN = [("qkv wgrad", M, 2304, 768), ("proj wgrad", M, 768, 768), ("fc1 wgrad", M, 3072, 768), ("fc2 wgrad", M, 768, 3072)]


def timeit(fn, reps):
    for _ in range(2):
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
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    with_db = not (len(sys.argv) > 3 and sys.argv[3] == "nodb")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    if which in ("nt", "all"):
        for name, m, n, k, epi in NT:
            a, b = rnd(m, k), rnd(n, k)
            bias = torch.randn(n, device=dev)
            res = rnd(m, n) if epi == _lib.EPI_BIAS_RES else None
            aux = rnd(m, n) if epi in (_lib.EPI_DGELU, _lib.EPI_MUL) else None
            out2 = torch.empty((m, n), dtype=torch.bfloat16, device=dev) if epi in (_lib.EPI_BIAS_GELU, _lib.EPI_BIAS_GELU_DG) else None
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            ms = timeit(lambda: ops.gemm_nt(a, b, epilogue=epi, bias=bias, res=res, aux=aux, out2=out2, out=out), reps)
            print(f"nt {name:10s} M={m} N={n:5d} K={k:5d}  {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)
    if which in ("tn", "all"):
        for name, m, n, k in TN:
            a, b = rnd(m, n), rnd(m, k)
            out = torch.empty((n, k), dtype=torch.float32, device=dev)
            db = torch.empty(n, dtype=torch.float32, device=dev)
            ms = timeit(lambda: ops.gemm_tn(a, b, out, dbias=db if with_db else None), reps)
            print(f"tn {name:10s} M={m} N={n:5d} K={k:5d}  {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
