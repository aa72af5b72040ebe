#!/usr/bin/env python3
"""Overlap summary of a rocprofv3 kernel trace of bench.py: per training step (delimited by the AdamW launches) the wall
time, the time with at least one / at least two kernels in flight, and per kernel family the summed launch durations by
HSA queue.  usage: summarize_trace.py <kernel_trace.csv> <out.md> [bench.json]"""
import collections
import csv
import re
import sys

src, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ends = [r["e"] for r in rows if "adamw" in r["Kernel_Name"]][1::2]      # two AdamW launches (weight-decay groups) end a step


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[-64:]


lines = ["# Kernel overlap in un-bracketed bench.py steps (rocprofv3 --kernel-trace)", ""]
if len(sys.argv) > 3:
    lines += ["bench line of the traced run: `" + open(sys.argv[3]).read().strip()[:400] + " …`", ""]
lines += ["| step | wall ms | >= 1 kernel in flight | >= 2 kernels in flight | idle |", "|---|---:|---:|---:|---:|"]
last = None
for k in range(1, len(ends) - 1):
    t0, t1 = ends[k], ends[k + 1]
    st = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
    ev = sorted([(r["s"], 1) for r in st] + [(r["e"], -1) for r in st])
    busy = over = depth = 0
    prev = None
    for t, d in ev:
        if prev is not None:
            busy += (t - prev) if depth >= 1 else 0
            over += (t - prev) if depth >= 2 else 0
        depth += d
        prev = t
    lines.append(f"| {k} | {(t1 - t0) / 1e6:.2f} | {busy / 1e6:.2f} | {over / 1e6:.2f} | {(t1 - t0 - busy) / 1e6:.2f} |")
    last = st
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    key = (short(r["Kernel_Name"]), r["Queue_Id"])
    agg[key][0] += r["e"] - r["s"]
    agg[key][1] += 1
lines += ["", "Last step, launch durations by kernel and queue (durations of overlapping launches add up to more than the wall time; a launch",
          "that shares the chip is stretched by the share it does not get):", "", "| kernel | queue | launches | total ms | avg us |", "|---|---|---:|---:|---:|"]
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    lines.append(f"| `{key[0]}` | {key[1]} | {v[1]} | {v[0] / 1e6:.2f} | {v[0] / v[1] / 1e3:.1f} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
