#!/usr/bin/env python3
"""HBM-side bytes per launch and kernel family from the two counter passes of tools/pmc_hbm.sh.

usage: python tools/summarize_pmc_hbm.py gpurun_out/pmc_hbm profiles/r3_pmc_hbm.md tools/pmc_hbm.json

Counter unit: KB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes, so
fetch bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores (MI355X_MICROARCH.md, HBM / rocprofv3
section).  Both count the memory side of L2, Infinity-Cache hits included: an upper bound on DRAM traffic."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

FAMILIES = [("gemm_nt", ("gemm_ntp_kernel", "gemm_nt_kernel")), ("gemm_tn", ("gemm_tnb_kernel", "gemm_tn_kernel")),
            ("attn_fwd", ("attn_fwd_kernel",)), ("attn_bwd", ("attn_bwd",)), ("ln_fwd", ("ln_fwd_kernel",)),
            ("ln_bwd", ("ln_bwd_kernel",)), ("adamw", ("adamw_kernel",))]
# algorithmic bytes per launch at ViT-B/16, 512 images (average over the family's launches of a step: operands read once,
# results written once; gemm_tn: both operands of the four weight gradients of a block, (620 + 310 + 775 + 775) / 4 MB)
ALGORITHMIC = {"gemm_nt": 814e6, "gemm_tn": 601e6, "attn_fwd": 620e6, "attn_bwd": 1240e6, "ln_fwd": 310e6, "ln_bwd": 725e6, "adamw": 1.3e9}


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return None


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(str(Path(d) / "**" / "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
            acc[fam][0] += 1
            acc[fam][1] += float(r["Counter_Value"]) * 1024.0
    return acc


def main():
    src, md, js = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
    f, w = load(src / "f", "FETCH_SIZE"), load(src / "w", "WRITE_SIZE")
    out = {}
    lines = ["# HBM-side traffic per kernel launch (rocprofv3 PMC)", "",
             "Two separate passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline` (ViT-B/16, 512 images / step),",
             "counters only (`tools/pmc_hbm.sh`; this table by `tools/summarize_pmc_hbm.py`).  fetch bytes = 2 x FETCH_SIZE (gfx950 tallies",
             "128-byte streaming requests at 64 bytes), WRITE_SIZE as counted; both are L2's memory side (Infinity-Cache hits included).", "",
             "| kernel family | launches | fetch MB / launch | write MB / launch | total MB / launch | algorithmic MB / launch | ratio |",
             "|---|---:|---:|---:|---:|---:|---:|"]
    for fam, _ in FAMILIES:
        if fam not in f or fam not in w:
            continue
        n = f[fam][0]
        fb, wb = 2.0 * f[fam][1] / n, w[fam][1] / max(w[fam][0], 1)
        tot = fb + wb
        out[fam] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "bytes_per_launch": round(tot),
                    "algorithmic_bytes_per_launch": ALGORITHMIC[fam],
                    "source": f"rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE passes, {md.name}"}
        lines.append(f"| {fam} | {n} | {fb / 1e6:.0f} | {wb / 1e6:.0f} | {tot / 1e6:.0f} | {ALGORITHMIC[fam] / 1e6:.0f} | {tot / ALGORITHMIC[fam]:.2f} |")
    md.write_text("\n".join(lines) + "\n")
    js.write_text(json.dumps(out, indent=1) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
