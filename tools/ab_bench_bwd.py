"""Interleaved A/B timing of fa_bwd for several builds of the library in ONE process (developer tool).

  python tools/ab_bench_bwd.py build/libA.so build/libB.so ... [--causal 1] [--rounds 3]
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--causal", type=int, default=0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()

torch.manual_seed(0)
q, k, v, do = (torch.randn(a.B, a.H, a.S, a.D, device="cuda").to(torch.bfloat16) for _ in range(4))
scale = a.D ** -0.5
libs = [fmod.load_library(p) for p in a.libs]
o, lse = fmod._fwd_raw(libs[0], q, k, v, bool(a.causal), scale, None, True)
times = {p: [] for p in a.libs}
ref = None
for p, lib in zip(a.libs, libs):
    g = fmod._bwd_raw(lib, q, k, v, o, lse, do, bool(a.causal), scale)
    torch.cuda.synchronize()
    if ref is None:
        ref = g
    else:
        print(f"{p}: max|g - g[first lib]| = " + " ".join(f"{(x.float() - y.float()).abs().max().item():.2e}" for x, y in zip(g, ref)))
for rnd in range(a.rounds):
    for p, lib in zip(a.libs, libs):
        for _ in range(2):
            fmod._bwd_raw(lib, q, k, v, o, lse, do, bool(a.causal), scale)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fmod._bwd_raw(lib, q, k, v, o, lse, do, bool(a.causal), scale)
        e1.record()
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / a.iters)
fl = attn_flops(a.B, a.H, a.S, a.D, bool(a.causal))
for p in a.libs:
    t = sorted(times[p])
    med = t[len(t) // 2]
    print(f"{p:44s} median {med:8.4f} ms  min {t[0]:8.4f} ms   (5-product algorithmic {2.5 * fl / med / 1e9:7.1f} TF/s)")
