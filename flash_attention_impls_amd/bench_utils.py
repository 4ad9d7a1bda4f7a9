"""Bench helpers mirroring the reference harness (code/triton_fa2/FA2-triton.py:249-268)."""
from __future__ import annotations

import statistics

import torch


def measure_latency(func, warmup: int = 10, iters: int = 100):
    """Per-iteration event timing with a sync after every call, like the reference's
    ``measure_latency`` (:249-268): returns mean/std (population)/min in ms."""
    for _ in range(warmup):
        func()
        torch.cuda.synchronize()
    times = []
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    for _ in range(iters):
        start.record()
        func()
        end.record()
        torch.cuda.synchronize()
        times.append(start.elapsed_time(end))
    return {"mean_ms": statistics.mean(times), "std_ms": statistics.pstdev(times),
            "min_ms": min(times), "iters": iters}


def attn_flops(B: int, H: int, S: int, D: int, causal: bool) -> float:
    """4*B*H*S^2*D (code/cutlass_cuda_fa1/run/test_flash_attn.cu:308), halved when causal
    (FlashAttention-2 convention)."""
    f = 4.0 * B * H * S * S * D
    return f / 2 if causal else f


def attn_bytes(B: int, H: int, S: int, D: int, in_bytes: int = 2, out_bytes: int = 2) -> float:
    """Q+K+V read once, O written once (test_flash_attn.cu:316-320), plus the fp32 LSE."""
    n = B * H * S * D
    return 3.0 * n * in_bytes + n * out_bytes + B * H * S * 4
