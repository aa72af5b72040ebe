#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the summary kept under profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof2 profiles/r1_bench 13 [bench.json]
       (13 = number of training steps in the profiled run, warm-up included)"""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path


def main():
    src, dst, steps = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
    bench = json.loads(Path(sys.argv[4]).read_text().strip().splitlines()[-1]) if len(sys.argv) > 4 else None
    stats = glob.glob(str(src / "**" / "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    dst.parent.mkdir(parents=True, exist_ok=True)
    shutil.copy(stats, str(dst) + "_kernel_stats.csv")
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        n = r["Name"]
        key = ("gemm_nt256_kernel" if "gemm_nt256" in n else "gemm_nt_kernel (128x128)" if "gemm_nt_kernel" in n else
               "gemm_tn_kernel" if "gemm_tn" in n else n.split("(")[0].replace("void ", "")[:60])
        f = fam.setdefault(key, [0, 0.0])
        f[0] += int(r["Calls"]); f[1] += float(r["TotalDurationNs"])
    lines = [f"# rocprofv3 kernel summary ({src.name}): `rocprofv3 --kernel-trace --stats -- python3 bench.py`", "",
             f"{steps} training steps profiled (ViT-B/16, 512 images / step / GPU); total kernel time {total / 1e6:.1f} ms "
             f"= {total / 1e6 / steps:.2f} ms per step.", ""]
    if bench:
        lines += [f"bench.py line of the same run: {bench['value']} images/s, {bench['ms_per_step']} ms/step; "
                  f"roofline (HIP events): gemm_nt {bench['roofline']['achieved']} TFLOP/s avg launch {bench['roofline']['avg_launch_ms']} ms, "
                  f"gemm_tn {bench['roofline']['also']['achieved']} TFLOP/s avg launch {bench['roofline']['also']['avg_launch_ms']} ms.", ""]
    lines += ["| kernel family | calls | total ms | ms / step | avg launch us | % |", "|---|---:|---:|---:|---:|---:|"]
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.0005:
            continue
        lines.append(f"| {k} | {c} | {t / 1e6:.2f} | {t / 1e6 / steps:.3f} | {t / 1e3 / c:.1f} | {100 * t / total:.1f} |")
    Path(str(dst) + "_summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
