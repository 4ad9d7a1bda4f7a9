"""Timing of the wide-head forward (head_dim 144 .. 256) on one GPU: TFLOP/s by the reference's FLOP model (4 B H S^2 D, halved
when causal).  usage: python tools/bench_wide.py [D ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from flash_attention_impls_amd import flash_attn
from flash_attention_impls_amd.bench_utils import time_launches

for D in [int(a) for a in sys.argv[1:]] or [256, 192]:
    for causal in (False, True):
        B, H, S = 4, 16, 4096
        q, k, v = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(3))
        r = time_launches(lambda: flash_attn(q, k, v, causal), warmup=5, iters=30)
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"D={D} causal={causal} (B,H,S)=({B},{H},{S}): {r['mean_ms']:.3f} ms  {fl / r['mean_ms'] / 1e9:.1f} TFLOP/s")
