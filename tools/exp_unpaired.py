"""Experiment: causal launches as query-block PAIRS (nqb + 1 block units per workgroup, the default at cfg3 / cfg4)
against SINGLE blocks, longest first, on an experiment build whose host heuristic can be overridden
(tools/build_variant.sh fwdexp -DFA_FWD_EXPERIMENTS; FA_MI355_FORCE_UNPAIRED=0/1), interleaved in one process.

  python tools/exp_unpaired.py [--lib build/libfwdexp.so] [--shapes 8,32,4096 1,16,16384]
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
ap.add_argument("--lib", default="build/libfwdexp.so")
ap.add_argument("--shapes", nargs="+", default=["8,32,4096", "1,16,16384", "4,32,8192", "16,32,2048"])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
fa_mod._lib_handle = fa_mod.load_library(a.lib)
for shp in a.shapes:
    B, H, S = (int(x) for x in shp.split(","))
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    outs, times = {}, {"0": [], "1": []}
    for mode in ("0", "1"):
        os.environ["FA_MI355_FORCE_UNPAIRED"] = mode
        outs[mode] = fa_mod.flash_attn(q, k, v, True)
    torch.cuda.synchronize()
    same = torch.equal(outs["0"], outs["1"])
    for _ in range(a.rounds):
        for mode in ("0", "1"):
            os.environ["FA_MI355_FORCE_UNPAIRED"] = mode
            for _ in range(3):
                fa_mod.flash_attn(q, k, v, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fa_mod.flash_attn(q, k, v, True)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / a.iters)
    fl = attn_flops(B, H, S, 128, True)
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    print(f"({B},{H},{S},128) causal: pairs {med['0']:.4f} ms ({fl / med['0'] / 1e9:.1f} TF) | single blocks, longest first {med['1']:.4f} ms "
          f"({fl / med['1'] / 1e9:.1f} TF) = {100 * (med['0'] / med['1'] - 1):+.1f} %   outputs bitwise equal: {same}")
