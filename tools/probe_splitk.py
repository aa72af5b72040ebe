#!/usr/bin/env python3
"""The left-over tile round of the persistent NT GEMM cut along K: time and error per shape with the cut off and with at most 2 / 3 / 4
parts per tile (VITK_NT_SPLITK, read by the library at every launch), the settings interleaved in one process.

    python tools/probe_splitk.py [reps]
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd import _lib, ops

# (name, kept samples, N, K, epilogue): token rows = kept * 197
CASES = [("fc2 fwd", 512, 768, 3072, _lib.EPI_BIAS_RES), ("fc2 fwd", 490, 768, 3072, _lib.EPI_BIAS_RES), ("fc2 fwd", 470, 768, 3072, _lib.EPI_BIAS_RES),
         ("fc2 fwd", 461, 768, 3072, _lib.EPI_BIAS_RES), ("dfc1", 461, 768, 3072, _lib.EPI_PLAIN), ("dqkv", 470, 768, 2304, _lib.EPI_PLAIN),
         ("proj", 461, 768, 768, _lib.EPI_BIAS_RES), ("qkv fwd", 461, 2304, 768, _lib.EPI_BIAS)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    for name, kept, n, k, epi in CASES:
        m = kept * 197
        a, b = rnd(m, k), rnd(n, k)
        bias = torch.randn(n, device=dev)
        res = rnd(m, n) if epi == _lib.EPI_BIAS_RES else None
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        run = lambda: ops.gemm_nt(a, b, epilogue=epi, bias=bias if epi != _lib.EPI_PLAIN else None, res=res, out=out)
        tiles = ((m + 255) // 256) * ((n + 255) // 256)
        line = f"{name:8s} kept={kept} N={n} K={k} tiles={tiles} ({tiles / 256:.2f} rounds, left {tiles % 256}):"
        ref = a[-4096:].float() @ b.float().t()              # the last rows: the tiles of the left-over round are among them
        if epi != _lib.EPI_PLAIN:
            ref += bias
        if res is not None:
            ref += res[-4096:].float()
        best = {}
        outs = {}
        for rnd_i in range(3):
            for mode in ("0", "2", "3", "4"):
                os.environ["VITK_NT_SPLITK"] = mode
                run()
                if rnd_i == 0:
                    outs[mode] = out.clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if not torch.equal(outs[mode], out):
                    line += f" [mode {mode}: NOT repeatable]"
                best[mode] = min(best.get(mode, 1e9), e0.elapsed_time(e1) / reps)
        for mode in ("0", "2", "3", "4"):
            err = ((outs[mode][-4096:].float() - ref).norm() / ref.norm()).item()
            diff = (outs[mode].float() - outs["0"].float()).abs().max().item()
            line += f"  s<={mode}: {best[mode] * 1e3:6.1f} us err {err:.2e} maxdiff {diff:.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
