#!/usr/bin/env python3
"""Per-instantiation SQ counters of the GEMM kernels from the counter passes of tools/pmc_gemm.sh.

usage: python tools/summarize_pmc_gemm.py gpurun_out/pmc_gemm profiles/r3_pmc_gemm.md

Every launch of tools/bench_gemm.py is one shape of the ViT-B/16 step (512 images); launches are keyed by kernel
instantiation and grid-independent shape order, counters are averaged per launch.  SQ cycle counters are summed over all
SIMDs of the chip (1 024)."""
import collections
import csv
import glob
import re
import sys
from pathlib import Path


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"([a-z_0-9]+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else re.sub(r"\(.*", "", name)


def main():
    src, md = Path(sys.argv[1]), Path(sys.argv[2])
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(str(src / "**" / "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "gemm" not in k:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k][0] += 1
                dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    counters = sorted({c for k in acc for c in acc[k]})
    kernels = sorted(acc)
    lines = ["# GEMM kernels, SQ counters per launch (rocprofv3 PMC)", "",
             "Counter-only passes over `python3 tools/bench_gemm.py all 2` (`tools/pmc_gemm.sh`), averaged per launch of each kernel",
             "instantiation; `us` is the launch duration under the counter pass (slower than an unprofiled launch).", "",
             "| counter | " + " | ".join(kernels) + " |", "|---|" + "---:|" * len(kernels)]
    lines.append("| launches / avg us | " + " | ".join(f"{dur[k][0]} / {dur[k][1] / max(dur[k][0], 1):.0f}" for k in kernels) + " |")
    for c in counters:
        lines.append(f"| {c} | " + " | ".join((f"{acc[k][c][1] / acc[k][c][0] / 1e6:.2f} M" if acc[k][c][0] else "-") for k in kernels) + " |")
    # derived: matrix-pipe busy share of the busy SIMD cycles, MFMA share of the wave cycles
    lines += ["", "Derived:", ""]
    for k in kernels:
        a = {c: acc[k][c][1] / acc[k][c][0] for c in acc[k] if acc[k][c][0]}
        parts = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("GRBM_GUI_ACTIVE"):
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
            us = dur[k][1] / max(dur[k][0], 1)
            parts.append(f"matrix pipe busy {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.0f} % of the launch's cycles on 1 024 SIMDs "
                         f"({a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_INSTS_MFMA']:.1f} cycles per MFMA; clock {cyc / us / 1e3:.2f} GHz)")
        if "SQ_WAIT_INST_ANY" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
            parts.append(f"issue-stalled {100 * a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:.0f} % of wave cycles")
        if "SQ_WAIT_ANY" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
            parts.append(f"parked {100 * a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.0f} %")
        if "SQ_WAIT_INST_LDS" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
            parts.append(f"waiting on LDS {100 * a['SQ_WAIT_INST_LDS'] / a['SQ_WAVE_CYCLES']:.0f} %")
        lines.append(f"* `{k}`: " + "; ".join(parts))
    md.write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
