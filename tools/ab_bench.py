"""Interleaved A/B timing of several builds of the library in ONE process (developer tool).

  python tools/ab_bench.py build/libA.so build/libB.so ... [--causal 1] [--rounds 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import importlib

fa_mod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp8"])
ap.add_argument("--zeros", type=int, default=0, help="1: all-zero inputs (the chip then holds its full clock: compares builds by cycles, not by energy)")
a = ap.parse_args()

torch.manual_seed(0)
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp8": torch.float8_e4m3fn}[a.dtype]
q, k, v = ((torch.zeros if a.zeros else torch.randn)(a.B, a.H, a.S, a.D, device="cuda").to(DT) for _ in range(3))
libs = [fa_mod.load_library(p) for p in a.libs]
ref = None
times = {p: [] for p in a.libs}


def run(lib):
    fa_mod._lib_handle = lib
    return fa_mod.flash_attn(q, k, v, bool(a.causal))


for p, lib in zip(a.libs, libs):
    o = run(lib)
    torch.cuda.synchronize()
    if ref is None:
        ref = o
    elif a.check:
        print(f"{p}: max|o - o[first lib]| = {(o.float() - ref.float()).abs().max().item():.3e}")
for rnd in range(a.rounds):
    for p, lib in zip(a.libs, libs):
        for _ in range(3):
            run(lib)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(lib)
        e1.record()
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / a.iters)
fl = attn_flops(a.B, a.H, a.S, a.D, bool(a.causal))
for p in a.libs:
    t = sorted(times[p])
    med = t[len(t) // 2]
    print(f"{p}: median {med:.4f} ms ({fl / med / 1e9:.1f} TF)  min {t[0]:.4f} ms ({fl / t[0] / 1e9:.1f} TF)")
