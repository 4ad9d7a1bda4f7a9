"""Tiny driver for rocprofv3: N launches of the fused forward at a named workload."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from flash_attention_impls_amd import flash_attn

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--bwd", type=int, default=0, help="1: time forward + backward steps (autograd path)")
a = ap.parse_args()
torch.manual_seed(0)
descale = None
if a.dtype == "fp8":
    f32 = [torch.randn(a.B, a.H, a.S, a.D, device="cuda") for _ in range(3)]
    descale = tuple(float(t.abs().max()) / 448.0 for t in f32)
    q, k, v = [(t / s).to(torch.float8_e4m3fn) for t, s in zip(f32, descale)]
else:
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    q, k, v = (torch.randn(a.B, a.H, a.S, a.D, device="cuda").to(dt) for _ in range(3))
if a.bwd:
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))
    do = torch.randn_like(q)
for _ in range(a.iters):
    o = flash_attn(q, k, v, bool(a.causal), descale=descale)
    if a.bwd:
        o.backward(do)
        q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
