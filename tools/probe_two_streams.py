#!/usr/bin/env python3
"""Probe: does running two half-batches concurrently on two streams (memory-bound kernels of one under the GEMMs of the other) beat
one full batch?  Forward + loss + backward only.  usage: python tools/probe_two_streams.py [steps]
VITK_GEMM_CUS=128 limits every GEMM launch to half of the compute units."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
from vision_transformers_torch_xla_amd.vit import create_model


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    crit = LabelSmoothingCrossEntropy(0.1)
    models = [create_model("vit_base_patch16_224", pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(dev).train()
              for _ in range(2)]
    g = torch.Generator(device=dev).manual_seed(0)
    x512 = torch.randn(512, 3, 224, 224, generator=g, device=dev)
    t512 = torch.randint(0, 1000, (512,), generator=g, device=dev)
    halves = [(x512[:256].contiguous(), t512[:256].contiguous()), (x512[256:].contiguous(), t512[256:].contiguous())]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def one_full():
        loss = crit(models[0](x512), t512)
        loss.backward()

    def two_halves(skew):
        losses = [None, None]
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                if i == 1 and skew:
                    torch.cuda._sleep(int(skew))
                losses[i] = crit(models[i](*halves[i][:1]), halves[i][1])
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                losses[i].backward()

    def two_halves_serial():
        for i in (0, 1):
            loss = crit(models[i](halves[i][0]), halves[i][1])
            loss.backward()

    def timeit(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    print(f"one batch of 512, one stream          {timeit(one_full, steps):7.2f} ms", flush=True)
    print(f"two batches of 256, one stream        {timeit(two_halves_serial, steps):7.2f} ms", flush=True)
    print(f"two batches of 256, two streams       {timeit(lambda: two_halves(0), steps):7.2f} ms", flush=True)
    print(f"two batches of 256, two streams, skew {timeit(lambda: two_halves(2000000), steps):7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
