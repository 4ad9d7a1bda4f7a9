"""Experiment (round 4, the tail of a launch): the last part of a causal launch of block pairs drawn from ONE pool of single blocks,
longest first, by a ticket counter (FwdParams::dyn_ctr) instead of being dealt out by block index, so that the XCDs that are ahead
take more of it.  Experiment build only (tools/build_variant.sh fwdexp -DFA_FWD_EXPERIMENTS; FA_MI355_DYN=1, FA_MI355_DYN_HEADS =
heads per XCD group in the pool, FA_MI355_DYN_SPARE = spare workgroups): one counter per process, one stream.

  python tools/exp_dyn_tail.py [--lib build/libfwdexp.so]
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
ap.add_argument("--shapes", nargs="+", default=["8,32,4096", "4,32,8192", "16,32,2048", "2,32,16384"])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
fa_mod._lib_handle = fa_mod.load_library(a.lib)
MODES = [("static", {"FA_MI355_DYN": "0"}), ("dyn 1/8", {"FA_MI355_DYN": "1"}), ("dyn 1 head", {"FA_MI355_DYN": "1", "FA_MI355_DYN_HEADS": "1"}),
         ("dyn 2 heads", {"FA_MI355_DYN": "1", "FA_MI355_DYN_HEADS": "2"}), ("dyn 6 heads", {"FA_MI355_DYN": "1", "FA_MI355_DYN_HEADS": "6"}),
         ("dyn 1/8, no spare", {"FA_MI355_DYN": "1", "FA_MI355_DYN_SPARE": "0"}), ("dyn 1/8, spare x2", {"FA_MI355_DYN": "1", "FA_MI355_DYN_SPARE": "512"})]


def setenv(env):
    for k in ("FA_MI355_DYN", "FA_MI355_DYN_HEADS", "FA_MI355_DYN_SPARE"):
        os.environ.pop(k, None)
    os.environ.update(env)


for shp in a.shapes:
    B, H, S = (int(x) for x in shp.split(","))
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    ref, times, same = None, {n: [] for n, _ in MODES}, {}
    for n, env in MODES:
        setenv(env)
        o, lse = fa_mod.flash_attn(q, k, v, True, return_lse=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = (o, lse)
        same[n] = torch.equal(o, ref[0]) and torch.equal(lse, ref[1])
    for _ in range(a.rounds):
        for n, env in MODES:
            setenv(env)
            for _ in range(3):
                fa_mod.flash_attn(q, k, v, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fa_mod.flash_attn(q, k, v, True)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / a.iters)
    fl = attn_flops(B, H, S, 128, True)
    base = sorted(times["static"])[len(times["static"]) // 2]
    print(f"({B},{H},{S},128) causal: " + " | ".join(
        f"{n}: {sorted(t)[len(t) // 2]:.4f} ms {fl / sorted(t)[len(t) // 2] / 1e9:.1f} TF ({100 * (base / sorted(t)[len(t) // 2] - 1):+.1f} %{'' if same[n] else ', OUTPUT DIFFERS'})"
        for n, t in times.items()), flush=True)
