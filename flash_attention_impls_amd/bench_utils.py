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


def try_max_batch(flash_attention, base_B: int = 1, H: int = 16, N: int = 1024, D: int = 32,
                  dtype=torch.float16, causal: bool = True, limit_B: int | None = None):
    """Largest batch whose forward + backward step fits in device memory, as the reference probes it
    (``try_max_batch``, FA2-triton.py:270-309): double the batch until an out-of-memory error, then binary
    search; one step = flash_attention(q,k,v) -> loss = o.float().pow(2).mean() -> backward.
    ``limit_B`` (not in the reference) bounds the search so that a shared box is not driven out of memory:
    the probe never allocates a batch above it and returns it when everything up to it fits."""
    device = "cuda"

    def fits(bsz: int) -> bool:
        try:
            q = torch.randn(bsz, H, N, D, device=device, dtype=dtype, requires_grad=True)
            k = torch.randn_like(q, requires_grad=True)
            v = torch.randn_like(q, requires_grad=True)
            o = flash_attention(q, k, v, causal=causal)
            loss = o.float().pow(2).mean()
            loss.backward()
            del q, k, v, o, loss
            torch.cuda.empty_cache()
            return True
        except RuntimeError as e:
            if "out of memory" in str(e).lower():
                torch.cuda.empty_cache()
                return False
            raise

    low, high = 1, base_B
    while True:                                   # grow until OOM (:274-289)
        if limit_B is not None and high > limit_B:
            high = limit_B + 1
            break
        if fits(high):
            low = high
            high *= 2
        else:
            break
    L, R, best = low, high - 1, low               # binary search in (low, high) (:290-308)
    while L <= R:
        mid = (L + R) // 2
        if fits(mid):
            best = mid
            L = mid + 1
        else:
            R = mid - 1
    return best
