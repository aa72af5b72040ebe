"""bench.py's launch contract: `--gpus N` runs N ranks - its own children when no launcher set WORLD_SIZE - or fails; it never
measures fewer ranks than it reports."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_mismatch_is_an_error():
    """a launcher that started another number of ranks than --gpus asks for: non-zero exit before anything is measured"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match --gpus 2" in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
def test_gpus_2_without_launcher_starts_two_ranks(device):
    """`python bench.py --gpus 2` alone: two ranks (sharing this box's one GPU, gradients averaged through gloo - RCCL refuses
    two ranks on one device), the whole multi-process flow - barriers, max-over-ranks time, rank 0's JSON line"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--model", "vit_tiny_patch16_224", "--no-cpu-baseline", "--prof-every", "0"],
                       env=_env(VITK_SHARE_GPU="1", VITK_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp2"
    assert rec["rccl_ranks"] == 0                          # the rehearsal backend is gloo, and the line says so
    assert rec["value"] > 0 and rec["scaling"] == "weak"
