#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the summary kept under profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof2 profiles/r1_bench 13 [bench.json] [workload]
       (13 = number of training steps in the profiled run, warm-up included; workload = vit_b_224 | vit_l_224 | vit_b_384 |
       deit_b_kd: adds the GEMM families' recomputed share of the bf16 MFMA peak)"""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path


# workload -> (title, width D, hidden F, blocks, tokens N, patches P, prefix tokens, images per step, teacher forward passes)
SHAPES = {"vit_b_224": ("ViT-B/16, 224 px, 512 images / step", 768, 3072, 12, 197, 196, 1, 512, 0),
          "vit_l_224": ("ViT-L/16, 224 px, 256 images / step", 1024, 4096, 24, 197, 196, 1, 256, 0),
          "vit_b_384": ("ViT-B/16, 384 px (577 tokens), 128 images / step", 768, 3072, 12, 577, 576, 1, 128, 0),
          "deit_b_kd": ("DeiT-B/16 distilled + ViT-B/16 teacher forward, 512 images / step", 768, 3072, 12, 198, 196, 2, 512, 1)}
PEAK = 2500.0       # dense bf16 MFMA, TFLOP/s


def gemm_tflop_per_step(w):
    """(NT family, TN family) TFLOP of one training step as EXECUTED: forward + data gradient in the NT kernels (the patch embedding
    has no data gradient), weight gradients in the TN kernels; stochastic depth 0.1 (linear over the blocks) runs a block's branches
    for 95 % of the samples on average; a distillation teacher adds one forward pass without stochastic depth"""
    _, D, F, L, N, P, npre, B, teacher = SHAPES[w]
    heads = 2 if npre == 2 else 1
    block = N * (4 * D * D + 2 * D * F)                    # MACs per image and block (qkv, proj, fc1, fc2)
    patch, head = P * 768 * D, heads * 1000 * D
    nt = 2 * B * (0.95 * 2 * L * block + patch + 2 * head)
    tn = 2 * B * (0.95 * L * block + patch + head)
    if teacher:
        nt += 2 * B * (L * 197 * (4 * D * D + 2 * D * F) + patch + 1000 * D)
    return nt / 1e12, tn / 1e12


def main():
    src, dst, steps = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
    bench = json.loads(Path(sys.argv[4]).read_text().strip().splitlines()[-1]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    workload = sys.argv[5] if len(sys.argv) > 5 else None
    stats = glob.glob(str(src / "**" / "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    dst.parent.mkdir(parents=True, exist_ok=True)
    shutil.copy(stats, str(dst) + "_kernel_stats.csv")
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        n = r["Name"]
        key = ("gemm_nt256_kernel" if "gemm_nt256" in n else "gemm_nt_kernel (128x128)" if "gemm_nt_kernel" in n else
               "gemm_tn_kernel" if "gemm_tn_kernel" in n else n.split("(")[0].replace("void ", "")[:60])
        f = fam.setdefault(key, [0, 0.0])
        f[0] += int(r["Calls"]); f[1] += float(r["TotalDurationNs"])
    lines = [f"# rocprofv3 kernel summary ({src.name}): `rocprofv3 --kernel-trace --stats -- python3 bench.py`", "",
             f"{steps} training steps profiled (ViT-B/16, 512 images / step / GPU); total kernel time {total / 1e6:.1f} ms "
             f"= {total / 1e6 / steps:.2f} ms per step.", ""]
    if workload:
        lines[0] = f"# rocprofv3 kernel summary ({src.name}): `rocprofv3 --kernel-trace --stats`, workload {workload}"
        lines[2] = lines[2].replace("ViT-B/16, 512 images / step / GPU", SHAPES[workload][0])
        nt_ms = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_nt" in r["Name"]) / 1e6 / steps
        tn_ms = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_tn" in r["Name"]) / 1e6 / steps
        nt, tn = gemm_tflop_per_step(workload)
        lines += [f"Recomputed from this table: the NT GEMM family (forward linears + data gradients) runs {nt:.2f} TFLOP per step in "
                  f"{nt_ms:.2f} ms = **{nt / nt_ms * 1e3:.0f} TFLOP/s = {nt / nt_ms * 1e3 / PEAK:.3f} of the {PEAK:.0f} TFLOP/s bf16 MFMA peak**; the TN family "
                  f"(weight gradients) {tn:.2f} TFLOP in {tn_ms:.2f} ms = {tn / tn_ms * 1e3:.0f} TFLOP/s = {tn / tn_ms * 1e3 / PEAK:.3f} "
                  f"(executed work: stochastic depth 0.1 runs a block's branches for 95 % of the samples on average).", ""]
    if bench and "roofline" in bench:
        lines += [f"bench.py line of the same run: {bench['value']} images/s, {bench['ms_per_step']} ms/step; "
                  f"roofline (HIP events): gemm_nt {bench['roofline']['achieved']} TFLOP/s avg launch {bench['roofline']['avg_launch_ms']} ms, "
                  f"gemm_tn {bench['roofline']['also']['achieved']} TFLOP/s avg launch {bench['roofline']['also']['avg_launch_ms']} ms.", ""]
    elif bench:
        lines += [f"tools/bench_configs.py line of the same (profiled) run: {bench['images_per_s']} images/s, {bench['ms_per_step']} ms/step.", ""]
    lines += ["| kernel family | calls | total ms | ms / step | avg launch us | % |", "|---|---:|---:|---:|---:|---:|"]
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.0005:
            continue
        lines.append(f"| {k} | {c} | {t / 1e6:.2f} | {t / 1e6 / steps:.3f} | {t / 1e3 / c:.1f} | {100 * t / total:.1f} |")
    Path(str(dst) + "_summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
