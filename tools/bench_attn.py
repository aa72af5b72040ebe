#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels on the ViT-B/16 shape (B=512, N=197, H=12): time and HBM rate.
usage: python tools/bench_attn.py [reps] [B] [N]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import ops


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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 197
    H = 12
    D = 64 * H
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(B * N, D, generator=g, device=dev).to(torch.bfloat16)
    ws = ops.Workspace()
    o, lse = ops.attention_fwd(qkv, B, N, H, 0.125)
    ms = timeit(lambda: ops.attention_fwd(qkv, B, N, H, 0.125), reps)
    byt = (qkv.numel() + o.numel()) * 2
    print(f"attention fwd  {ms * 1e3:8.1f} us   {byt / ms / 1e9:6.2f} TB/s   {4.0 * B * H * N * N * 64 / ms / 1e9:6.1f} TFLOP/s", flush=True)
    ms = timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, N, H, 0.125, ws=ws), reps)
    byt = (2 * qkv.numel() + 2 * o.numel()) * 2
    print(f"attention bwd  {ms * 1e3:8.1f} us   {byt / ms / 1e9:6.2f} TB/s   {10.0 * B * H * N * N * 64 / ms / 1e9:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
