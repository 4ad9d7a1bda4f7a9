"""Backward timing on the box: fa_bwd (pre-pass + dK/dV kernel writing dS + dQ GEMM; FA_MI355_BWD_DS=0: pre-pass + dQ kernel + dK/dV
kernel, recompute) per BASELINE shape, and the
reference harness' fwd+bwd step (FA2-triton.py:357-372: out = flash_attention(q,k,v); out.sum().backward()).
Algorithmic FLOPs: 2.5 x forward (five S x S x D products; the two-kernel form executes seven)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import flash_attention_impls_amd as fa
import importlib

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")   # the module, not the function of the same name
from flash_attention_impls_amd.bench_utils import attn_flops, measure_latency

CONFIGS = {
    "cfg2": (4, 8, 1024, 64, torch.bfloat16, False),
    "cfg3": (8, 32, 4096, 128, torch.bfloat16, True),
    "cfg3nc": (8, 32, 4096, 128, torch.bfloat16, False),
    "cfg3fp16": (8, 32, 4096, 128, torch.float16, True),
    "cfg4": (1, 16, 16384, 128, torch.bfloat16, True),
    "ref-bwd": (1, 16, 1024, 64, torch.float16, True),
    "d64": (8, 32, 4096, 64, torch.bfloat16, True),
    "d64nc": (8, 32, 4096, 64, torch.bfloat16, False),
    "ref-main": (1, 16, 1024, 32, torch.float16, True),
    "d256": (4, 16, 4096, 256, torch.bfloat16, True),          # wide heads: fa_bwd_wide_ds + three library GEMMs
    "d256nc": (4, 16, 4096, 256, torch.bfloat16, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,cfg3,cfg3nc,cfg4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    lib = fa.load_library(args.lib)
    print(torch.cuda.get_device_name(0))
    print(f"{'config':10s} {'shape':22s} {'dtype':9s} {'causal':6s} {'bwd ms':>8s} {'bwd TF/s':>9s} {'fwd ms':>8s} "
          f"{'fwd+bwd step ms':>16s} {'step TF/s':>10s}", flush=True)
    for name in args.configs.split(","):
        B, H, S, D, dt, causal = CONFIGS[name]
        torch.manual_seed(0)
        q, k, v, do = [torch.randn(B, H, S, D, device="cuda").to(dt) for _ in range(4)]
        scale = D ** -0.5
        o, lse = fmod._fwd_raw(lib, q, k, v, causal, scale, None, True)
        bwd = fmod._bwd_wide if D > 128 else fmod._bwd_raw
        mb = measure_latency(lambda: bwd(lib, q, k, v, o, lse, do, causal, scale), warmup=3, iters=args.iters)
        mf = measure_latency(lambda: fmod._fwd_raw(lib, q, k, v, causal, scale, None, True), warmup=3, iters=args.iters)
        qg, kg, vg = [t.clone().requires_grad_(True) for t in (q, k, v)]

        def step():
            qg.grad = kg.grad = vg.grad = None
            fa.flash_attn(qg, kg, vg, causal).sum().backward()
        ms = measure_latency(step, warmup=3, iters=args.iters)
        fl = attn_flops(B, H, S, D, causal)
        print(f"{name:10s} {str((B, H, S, D)):22s} {str(dt)[6:]:9s} {str(causal):6s} {mb['mean_ms']:8.3f} "
              f"{2.5 * fl / mb['mean_ms'] / 1e9:9.1f} {mf['mean_ms']:8.3f} {ms['mean_ms']:16.3f} "
              f"{3.5 * fl / ms['mean_ms'] / 1e9:10.1f}", flush=True)


if __name__ == "__main__":
    main()
