"""Quick on-box sanity + timing sweep (developer tool, not part of the test suite).

  python tools/gpu_check.py [--quick]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch
import torch.nn.functional as F

from flash_attention_impls_amd import flash_attn
from flash_attention_impls_amd.bench_utils import attn_flops


def ref_sdpa(q, k, v, causal):
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal,
                                          scale=1.0 / math.sqrt(q.shape[-1]))


def check(B, H, S, D, dtype, causal, seed=0, mul=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = (torch.randn(B, H, S, D, device="cuda", generator=g) * mul).to(dtype)
    k = (torch.randn(B, H, S, D, device="cuda", generator=g) * mul).to(dtype)
    v = (torch.randn(B, H, S, D, device="cuda", generator=g) * mul).to(dtype)
    o, lse = flash_attn(q, k, v, causal, return_lse=True)
    torch.cuda.synchronize()
    ref = ref_sdpa(q, k, v, causal)
    err = (o.float() - ref).abs().max().item()
    # lse reference
    s = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) / math.sqrt(D)
    if causal:
        i = torch.arange(S, device="cuda")[:, None]
        j = torch.arange(S, device="cuda")[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    lse_ref = torch.logsumexp(s, dim=-1)
    lerr = (lse - lse_ref).abs().max().item()
    nan = bool(torch.isnan(o.float()).any())
    print(f"check B={B} H={H} S={S} D={D} {str(dtype)[6:]} causal={int(causal)} mul={mul}: "
          f"max|o-ref|={err:.3e} max|lse-ref|={lerr:.3e} nan={nan}", flush=True)
    return err


def timeit(B, H, S, D, dtype, causal, iters=20, warmup=5):
    q = torch.randn(B, H, S, D, device="cuda").to(dtype)
    k = torch.randn(B, H, S, D, device="cuda").to(dtype)
    v = torch.randn(B, H, S, D, device="cuda").to(dtype)
    for _ in range(warmup):
        flash_attn(q, k, v, causal)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        flash_attn(q, k, v, causal)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = attn_flops(B, H, S, D, causal) / (ms * 1e-3) / 1e12
    print(f"time  B={B} H={H} S={S} D={D} {str(dtype)[6:]} causal={int(causal)}: {ms:.4f} ms  {tf:.1f} TFLOP/s "
          f"({100 * tf / 2516.6:.1f}% of bf16 MFMA peak)", flush=True)
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--lib", default=None, help="alternative build of the library to check")
    args = ap.parse_args()
    if args.lib:
        import importlib
        fa_mod = importlib.import_module("flash_attention_impls_amd.flash_attn")
        fa_mod._lib_handle = fa_mod.load_library(args.lib)
    print(torch.cuda.get_device_name(0), flush=True)
    bf, hf = torch.bfloat16, torch.float16
    for (B, H, S, D, dt, c) in [
        (1, 1, 64, 128, bf, False), (1, 1, 64, 128, bf, True),
        (1, 2, 256, 128, bf, False), (1, 2, 256, 128, bf, True),
        (2, 2, 512, 128, bf, True), (1, 2, 200, 128, bf, True), (1, 3, 77, 64, bf, False),
        (1, 1, 1, 128, bf, True), (1, 1, 513, 128, bf, True), (1, 2, 1024, 64, bf, False),
        (1, 2, 1024, 64, hf, True), (1, 2, 320, 128, hf, False), (2, 3, 1000, 128, bf, True),
        (1, 9, 300, 64, hf, True),
    ]:
        check(B, H, S, D, dt, c)
    check(1, 1, 64, 64, hf, False, seed=3, mul=8.0)
    check(1, 2, 2048, 128, bf, True, seed=5, mul=3.0)
    if not args.quick:
        timeit(4, 8, 1024, 64, bf, False)
        timeit(8, 32, 4096, 128, bf, True)
        timeit(8, 32, 4096, 128, bf, False)
        timeit(1, 16, 16384, 128, bf, True)
        timeit(8, 32, 4096, 128, hf, True)
        timeit(8, 32, 4096, 64, bf, True)


if __name__ == "__main__":
    main()
