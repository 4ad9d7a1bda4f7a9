"""Forward and backward latency with grouped key/value heads next to the equal-head-count case.

    python tools/bench_gqa.py [B H S D] [--hkv 8 4 1] [--sk N_k]

--sk gives K, V a length of their own (S queries against N_k keys; the causal mask is bottom-right aligned).

Same timing rules as the reference's measure_latency (FA2-triton.py:249-268: warm-ups, then an event pair per iteration).
FLOPs are those of the H query heads (4 B H D x visible (query, key) pairs; backward 2.5x).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_attention_impls_amd as fa  # noqa: E402
import importlib  # noqa: E402

hostmod = importlib.import_module("flash_attention_impls_amd.flash_attn")


def timed(fn, warmup=10, iters=50):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[8, 32, 4096, 128])
    ap.add_argument("--hkv", nargs="*", type=int, default=[32, 8, 4, 1])
    ap.add_argument("--sk", type=int, default=None)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    B, H, S, D = a.shape
    Sk = a.sk or S
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    lib = fa.load_library()
    torch.manual_seed(0)
    q = torch.randn(B, H, S, D, device="cuda").to(dt)
    do = torch.randn(B, H, S, D, device="cuda").to(dt)
    for causal in (True, False):
        if causal and Sk < S:
            continue
        pairs = S * Sk - (S * (S - 1) / 2.0 if causal else 0.0)          # visible (query, key) pairs per head
        if Sk == S and causal:
            pairs = S * S / 2.0                                           # the FlashAttention-2 convention of bench.py
        fl = 4.0 * B * H * D * pairs
        for hkv in a.hkv:
            k = torch.randn(B, hkv, Sk, D, device="cuda").to(dt)
            v = torch.randn(B, hkv, Sk, D, device="cuda").to(dt)
            scale = D ** -0.5
            o, lse = hostmod._fwd_raw(lib, q, k, v, causal, scale, None, True)
            t_f = timed(lambda: hostmod._fwd_raw(lib, q, k, v, causal, scale, None, True))
            t_b = timed(lambda: hostmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale))
            print(f"({B},{H},{S}x{Sk},{D}) {a.dtype} causal={int(causal)} H_kv={hkv:3d}: fwd {t_f:7.3f} ms {fl / t_f / 1e9:7.1f} TFLOP/s"
                  f" | bwd {t_b:7.3f} ms {2.5 * fl / t_b / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
