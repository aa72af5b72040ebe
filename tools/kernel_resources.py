#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/spill/occupancy per kernel of the libvitk sources (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "vision_transformers_torch_xla_amd" / "csrc"


def main(names):
    names = names or [p.stem for p in sorted(CSRC.glob("*.hip"))]
    for n in names:
        src = CSRC / f"{n}.hip"
        extra = {"gemm": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"], "gemm_ntq": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"], "attention": ["-fno-slp-vectorize"]}.get(n, [])      # as build.py's EXTRA_FLAGS
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", *extra, "-c", str(src), "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        cur = {}
        rows = []
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
            if not m:
                continue
            txt = m.group(1).strip()
            if txt.startswith("Function Name:"):
                if cur:
                    rows.append(cur)
                cur = {"name": txt.split(":", 1)[1].strip()}
            elif ":" in txt:
                k, v = txt.split(":", 1)
                cur[k.strip()] = v.strip()
        if cur:
            rows.append(cur)
        print(f"== {n}.hip")
        for c in rows:
            dem = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            print(f"  {dem:48s} vgpr={c.get('VGPRs','?'):>4} agpr={c.get('AGPRs','?'):>3} sgpr={c.get('TotalSGPRs','?'):>3} "
                  f"spill={c.get('VGPR Spill','?')}/{c.get('SGPR Spill','?')} scratch={c.get('ScratchSize [bytes/lane]','?')} "
                  f"lds={c.get('LDS Size [bytes/block]','?'):>6} occ={c.get('Occupancy [waves/SIMD]','?')}")


if __name__ == "__main__":
    main(sys.argv[1:])
