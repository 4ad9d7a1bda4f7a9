"""The reference's `__main__` flow (code/triton_fa2/FA2-triton.py:328-376) on this library's drop-in entry point:
correctness against SDPA, forward latency, forward + backward latency with the reference's loss, max-batch probe --
same shape (B=1, H=16, N=1024, D=32, fp16, causal), same warm-up / iteration counts, same report lines.

    python tools/reference_harness.py [--max-batch-limit 4096] [--shape B H N D]

--max-batch-limit bounds the memory probe (the reference doubles the batch until the device runs out of memory; on a
shared box with 288 GB that is neither quick nor polite): the probe stops at the limit and says so.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_impls_amd import flash_attention  # noqa: E402  (the reference's spelling of the entry point)
from flash_attention_impls_amd.bench_utils import measure_latency, try_max_batch  # noqa: E402


def sdpa_reference(q, k, v, causal):
    """fp32 SDPA with scale 1/sqrt(D) -- what the reference checks itself against (FA2-triton.py:311-323)."""
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal).to(q.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", nargs=4, type=int, default=[1, 16, 1024, 32])
    ap.add_argument("--max-batch-limit", type=int, default=4096)
    ap.add_argument("--no-causal", action="store_true")
    a = ap.parse_args()
    B, H, N, D = a.shape
    causal, dtype = not a.no_causal, torch.float16
    torch.manual_seed(0)
    q, k, v = [torch.randn(B, H, N, D, device="cuda", dtype=dtype, requires_grad=True) for _ in range(3)]

    with torch.no_grad():
        ref = sdpa_reference(q, k, v, causal)
        our = flash_attention(q, k, v, causal=causal)
        max_abs = (our - ref).abs().max().item()
    print(f"[Correctness] max_abs_diff_vs_sdpa: {max_abs:.4e}")

    tokens = B * H * N
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    f = measure_latency(lambda: flash_attention(q, k, v, causal=causal), warmup=10, iters=100)
    print(f"[Latency|Forward] mean={f['mean_ms']:.3f} ms (std={f['std_ms']:.3f}, iters={f['iters']}), "
          f"throughput={tokens / (f['mean_ms'] / 1e3):.1f} tokens/s, peak_mem={torch.cuda.max_memory_allocated() / 1e6:.1f} MB")

    def fwd_bwd():
        o = flash_attention(q, k, v, causal=causal)
        o.float().pow(2).mean().backward()
        for t in (q, k, v):
            if t.grad is not None:
                t.grad.zero_()
        return o

    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    e = measure_latency(fwd_bwd, warmup=10, iters=50)
    print(f"[Latency|Fwd+Bwd] mean={e['mean_ms']:.3f} ms (std={e['std_ms']:.3f}, iters={e['iters']}), "
          f"throughput={tokens / (e['mean_ms'] / 1e3):.1f} tokens/s, peak_mem={torch.cuda.max_memory_allocated() / 1e6:.1f} MB")

    max_b = try_max_batch(flash_attention, base_B=1, H=H, N=N, D=D, dtype=dtype, causal=causal, limit_B=a.max_batch_limit)
    note = f" (probe bounded at {a.max_batch_limit})" if max_b >= a.max_batch_limit else ""
    print(f"[MaxBatch] max_batch_size at (H={H}, N={N}, D={D}, dtype={dtype}): {max_b}{note}")


if __name__ == "__main__":
    main()
