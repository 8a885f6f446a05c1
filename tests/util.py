"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import torch


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# tolerances, stated once (rel-L2 against the fp32 oracle on identical bf16-rounded inputs/weights):
TOL_F32_OUT = 1e-3     # fp32-accumulate kernels with fp32 output
TOL_BF16_OUT = 3e-3    # same + one bf16 output rounding (floor 2^-9/sqrt(3) = 1.1e-3) and bf16 P in attention


def bf16_round(t):
    return t.bfloat16().float()


def sd_to(sd, dtype):
    return {k: (v.to(dtype) if torch.is_floating_point(v) else v) for k, v in sd.items()}
