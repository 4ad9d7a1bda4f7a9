"""Experiment: few-head, long-sequence launches -- the per-GPU shards of a strong split (cfg4: 16 heads of
S = 16384 over 8 GPUs = 2 heads) -- on 256-row workgroups (the default: 8 waves, two per SIMD) against 128-row workgroups of the
32x32x16 kernel (4 waves, one per SIMD, twice the workgroups), pairs against single blocks, on an experiment build whose host
rules can be overridden (tools/build_variant.sh fwdexp -DFA_FWD_EXPERIMENTS; FA_MI355_ROWS, FA_MI355_FORCE_UNPAIRED).

  python tools/exp_rows.py [--lib build/libfwdexp.so]
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
ap.add_argument("--shapes", nargs="+", default=["1,1,16384", "1,2,16384", "1,4,16384", "1,8,16384", "1,16,16384", "1,4,4096", "2,4,8192"])
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
fa_mod._lib_handle = fa_mod.load_library(a.lib)
MODES = [("256 rows, host rule", {"FA_MI355_ROWS": "2"}), ("256 rows, pairs", {"FA_MI355_ROWS": "2", "FA_MI355_FORCE_UNPAIRED": "0"}),
         ("256 rows, single", {"FA_MI355_ROWS": "2", "FA_MI355_FORCE_UNPAIRED": "1"}),
         ("128 rows, pairs", {"FA_MI355_ROWS": "1", "FA_MI355_FORCE_UNPAIRED": "0"}), ("128 rows, single", {"FA_MI355_ROWS": "1", "FA_MI355_FORCE_UNPAIRED": "1"})]
for shp in a.shapes:
    B, H, S = (int(x) for x in shp.split(","))
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    for causal in (True, False):
        ref, row = None, []
        for name, env in MODES:
            if not causal and "FA_MI355_FORCE_UNPAIRED" in env:
                if env["FA_MI355_FORCE_UNPAIRED"] == "1":
                    continue
            for kk in ("FA_MI355_ROWS", "FA_MI355_FORCE_UNPAIRED"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            o = fa_mod.flash_attn(q, k, v, causal)
            torch.cuda.synchronize()
            if ref is None:
                ref = o
            err = float((o.float() - ref.float()).abs().max())
            for _ in range(5):
                fa_mod.flash_attn(q, k, v, causal)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fa_mod.flash_attn(q, k, v, causal)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.iters)
            ms = sorted(ts)[1]
            row.append(f"{name}: {ms:.4f} ms {attn_flops(B, H, S, 128, causal) / ms / 1e9:7.1f} TF (|d| {err:.1e})")
        print(f"({B},{H},{S},128) {'causal' if causal else 'non-causal'}:  " + "  |  ".join(row), flush=True)
