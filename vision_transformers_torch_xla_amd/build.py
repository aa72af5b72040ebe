"""Builds libvitk.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in a CPU-only container; the resulting
shared object sits next to the sources (``csrc/libvitk.so``) and is what ``_lib.py`` loads through ctypes.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libvitk.so"
SOURCES = ["api.hip", "layernorm.hip", "gemm.hip", "gemm_ntq.hip", "attention.hip", "elementwise.hip", "loss.hip", "optim.hip", "preprocess.hip"]
HEADERS = [CSRC / "common.h", CSRC / "gemm_nt.h", CSRC.parent.parent / "include" / "vitk.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-Wno-cuda-compat", "-Wno-pass-failed"]
# gemm.hip: the tile claim of the persistent kernel is ONE lane's atomic whose result is used a whole tile later; hipcc's atomic
# optimizer would turn it into a wave-wide reduction and wait for the result on the spot
EXTRA_FLAGS = {"gemm.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
               "gemm_ntq.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
               # beside MFMAs a packed fp32 instruction costs more issue time than the two scalar ones it replaces; the
               # softmax arithmetic between the attention products is faster left unpacked (forward -2 %, backward -1..3 %)
               "attention.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvitk.so cannot be built on this machine")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> Path:
    """Compile every .hip source to an object file (in parallel) and link libvitk.so.  Returns the library path."""
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = CSRC / src
        o = objdir / (s.stem + ".o")
        if force or _stale(o, [s, *HEADERS]):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s.name, []), "-c", str(s), "-o", str(o)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [objdir / (Path(s).stem + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
