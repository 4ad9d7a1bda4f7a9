"""Where the dS hand-off stops paying at head_dim <= 64: recompute (build/librecompute.so) against a build that takes the hand-off
wherever it qualifies (build/libds_always.so, -DFA_BWD_DS_ALWAYS), over dS images from 32 MiB to 4 GiB (developer tool)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

os.environ["FA_MI355_BWD_DS_MAX_GIB"] = "32"
arms = {"recompute": fmod.load_library(os.path.join("build", "librecompute.so")), "hand-off": fmod.load_library(os.path.join("build", "libds_always.so"))}
print(torch.cuda.get_device_name(0))
print(f"{'(B,H,S,D)':24s} {'causal':6s} {'dS image MiB':>12s} {'recompute ms':>13s} {'hand-off ms':>12s} {'gain':>7s}")
SHAPES = [(B, H, S, D, c) for D in (64, 32) for (B, H, S) in ((1, 8, 1024), (2, 16, 1024), (4, 16, 1024), (8, 16, 1024), (8, 32, 1024), (16, 32, 1024),
                                                                (1, 8, 2048), (2, 16, 2048), (4, 16, 2048), (8, 32, 2048), (1, 8, 4096), (1, 32, 4096), (4, 32, 4096))
          for c in (True, False)]
for B, H, S, D, causal in SHAPES:
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(4))
    scale = D ** -0.5
    o, lse = fmod._fwd_raw(arms["hand-off"], q, k, v, causal, scale, None, True)
    times = {a: [] for a in arms}
    fmod._plan_cache.clear()
    for rnd in range(3):
        for a, lib in arms.items():
            fmod._plan_cache.clear()           # (the plan is cached per shape, not per library)
            for _ in range(3):
                fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 30 if B * H * S <= 2 ** 17 else 10
            e0.record()
            for _ in range(iters):
                fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
            e1.record()
            torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / iters)
    t0, t1 = (sorted(times[a])[1] for a in ("recompute", "hand-off"))
    img = 2.0 * B * H * S * S / 2 ** 20
    print(f"{str((B, H, S, D)):24s} {str(causal):6s} {img:12.0f} {t0:13.4f} {t1:12.4f} {100 * (t0 / t1 - 1):6.1f}%", flush=True)
