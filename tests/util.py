"""Shared helpers for the parity tests."""
import torch

# Floating-point bar from the north star: HIP results within 1e-3 relative of the CPU oracle evaluated with the
# same bf16 storage points (oracle `emulate=True`), measured as relative L2 error per tensor.
REL_TOL = 1e-3


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    den = b.norm().item()
    if den == 0.0:
        return a.norm().item()
    return (a - b).norm().item() / den


def bf16_excess(a: torch.Tensor, ref: torch.Tensor, atol: float = 1e-5) -> float:
    """How far a bf16 result is from being a correct rounding of the fp32/fp64 reference, in units of the allowance.

    A correctly rounded bf16 value lies within 2^-8 |ref| of ref (half an ulp, worst case); `atol` absorbs the fp32
    accumulation noise of the kernel for results that cancel to (almost) zero, where a bf16 ulp is arbitrarily small.
    Returns max |a-ref| / (2^-8 |ref| * 1.02 + atol); <= 1 means every element is within one rounding of the reference."""
    a = a.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    allow = ref.abs() * (2.0 ** -8) * 1.02 + atol
    return float(((a - ref).abs() / allow).max().item())


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------------------------------------------------------
# The independent anchor.  The bf16-emulating oracle states the kernels' own rounding policy, so a commit that moves a
# rounding point moves the kernel and that oracle together.  The fp32 oracle (emulate=False, exact math) does not move:
# tests/golden/fp32_distance.json holds, per comparison, the distance to it that was measured when the file was last
# recorded, and a later run may not exceed 1.25 x that value.  Recording: run the GPU suite with
# VITK_RECORD_FP32_DISTANCE=<file> and commit the file as tests/golden/fp32_distance.json (the diff of that file then
# shows, entry by entry, what a change did to the distance from exact math).
import json as _json
import os as _os
from pathlib import Path as _Path

FP32_ANCHOR_FILE = _Path(__file__).resolve().parent / "golden" / "fp32_distance.json"
FP32_ANCHOR_SLACK = 1.25
_anchor_cache = {}


def _anchor_table():
    if "t" not in _anchor_cache:
        _anchor_cache["t"] = _json.loads(FP32_ANCHOR_FILE.read_text()) if FP32_ANCHOR_FILE.exists() else {}
    return _anchor_cache["t"]


def fp32_anchor(name: str, value: float, floor: float = 1e-7) -> float:
    """`value` = a measured relative distance between a HIP result and the fp32 (non-emulating) oracle for comparison `name`.
    Check mode (default): it must not exceed FP32_ANCHOR_SLACK x the recorded value (+ floor, for distances that are
    rounding residue of exact results).  Record mode (VITK_RECORD_FP32_DISTANCE=<file>): the value is written to that file."""
    value = float(value)
    rec = _os.environ.get("VITK_RECORD_FP32_DISTANCE")
    if rec:
        path = _Path(rec)
        table = _json.loads(path.read_text()) if path.exists() else {}
        table[name] = float(f"{value:.4e}")
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(".tmp")
        tmp.write_text(_json.dumps(table, indent=1, sort_keys=True) + "\n")
        tmp.replace(path)
        return value
    table = _anchor_table()
    assert name in table, f"no recorded fp32-oracle distance for {name!r}: record tests/golden/fp32_distance.json (see tests/util.py)"
    bound = FP32_ANCHOR_SLACK * table[name] + floor
    assert value <= bound, f"{name}: distance to the fp32 oracle {value:.3e} exceeds {FP32_ANCHOR_SLACK} x recorded {table[name]:.3e}"
    return value
