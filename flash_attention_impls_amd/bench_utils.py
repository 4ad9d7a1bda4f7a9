"""Bench helpers: this build's event-timing loop, the FLOP / byte model and the reference harness' max-batch probe."""
from __future__ import annotations

import statistics

import torch


def time_launches(func, warmup: int = 10, iters: int = 100, sync_each: bool = False):
    """GPU time of `iters` calls of `func`, one HIP event between consecutive calls on the current stream (iters + 1
    events, read back after ONE final synchronisation, so the stream stays full and a call's host cost hides behind the
    previous call's kernel).  Returns mean / population std / min / median / max in ms.
    `sync_each=True` drains the stream after every call instead -- the reference harness' rule (`measure_latency`,
    code/triton_fa2/FA2-triton.py:249-268: a sync per iteration), which adds the launch latency of an idle stream to each
    sample; the reports that quote the reference's semantics use it."""
    for _ in range(warmup):
        func()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    if sync_each:
        times = []
        for i in range(iters):
            marks[i].record()
            func()
            marks[i + 1].record()
            marks[i + 1].synchronize()
            times.append(marks[i].elapsed_time(marks[i + 1]))
    else:
        marks[0].record()
        for i in range(iters):
            func()
            marks[i + 1].record()
        torch.cuda.synchronize()
        times = [marks[i].elapsed_time(marks[i + 1]) for i in range(iters)]
    ts = sorted(times)
    return {"mean_ms": statistics.fmean(times), "std_ms": statistics.pstdev(times), "min_ms": ts[0],
            "median_ms": ts[len(ts) // 2], "max_ms": ts[-1], "iters": iters}


def measure_latency(func, warmup: int = 10, iters: int = 100):
    """The reference harness' name and call shape (FA2-triton.py:249): its per-iteration-sync rule on this build's timing loop."""
    return time_launches(func, warmup, iters, sync_each=True)


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
