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
