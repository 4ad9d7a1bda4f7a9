"""Backward: dS hand-off (build/libds.so or the in-tree library) against the recompute path (-DFA_BWD_DS_DISABLE build) over a
list of shapes, both arms interleaved in one process (developer tool; the table in profiles/ comes from here).

  bash tools/build_variant.sh recompute -DFA_BWD_DS_DISABLE && python tools/sweep_bwd_handoff.py
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

SHAPES = [  # name, B, H, Hkv, S, D, dtype, causal
    ("cfg2", 4, 8, 8, 1024, 64, torch.bfloat16, False), ("ref-bwd", 1, 16, 16, 1024, 64, torch.float16, True),
    ("ref-main", 1, 16, 16, 1024, 32, torch.float16, True), ("small128", 2, 8, 8, 2048, 128, torch.bfloat16, True),
    ("d64", 8, 32, 32, 4096, 64, torch.bfloat16, True), ("d64nc", 8, 32, 32, 4096, 64, torch.bfloat16, False),
    ("cfg3", 8, 32, 32, 4096, 128, torch.bfloat16, True), ("cfg3nc", 8, 32, 32, 4096, 128, torch.bfloat16, False),
    ("cfg3fp16", 8, 32, 32, 4096, 128, torch.float16, True), ("cfg3gqa8", 8, 32, 8, 4096, 128, torch.bfloat16, True),
    ("cfg4", 1, 16, 16, 16384, 128, torch.bfloat16, True), ("s8k", 2, 32, 32, 8192, 128, torch.bfloat16, True),
    ("s2k", 32, 32, 32, 2048, 128, torch.bfloat16, True), ("s1k", 64, 32, 32, 1024, 128, torch.bfloat16, True),
    ("s512", 128, 32, 32, 512, 128, torch.bfloat16, True),
]
arms = {"recompute": fmod.load_library(os.path.join("build", "librecompute.so")), "hand-off": fmod.load_library()}
print(torch.cuda.get_device_name(0))
print(f"{'shape':10s} {'(B,H,Hkv,S,D)':26s} {'dtype':9s} {'causal':6s} {'recompute ms':>13s} {'hand-off ms':>12s} {'gain':>7s} {'hand-off TF/s':>14s} {'dS GiB':>7s} equal")
for name, B, H, Hkv, S, D, dt, causal in SHAPES:
    torch.manual_seed(0)
    q, do = (torch.randn(B, H, S, D, device="cuda").to(dt) for _ in range(2))
    k, v = (torch.randn(B, Hkv, S, D, device="cuda").to(dt) for _ in range(2))
    scale = D ** -0.5
    o, lse = fmod._fwd_raw(arms["hand-off"], q, k, v, causal, scale, None, True)
    grads, times = {}, {a: [] for a in arms}
    for a, lib in arms.items():
        grads[a] = fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
    torch.cuda.synchronize()
    same = all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(grads["recompute"], grads["hand-off"]))
    iters = 20 if S * B * H <= 2 ** 18 else 8
    for rnd in range(3):
        for a, lib in arms.items():
            for _ in range(2):
                fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
            e1.record()
            torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / iters)
    t0, t1 = (sorted(times[a])[1] for a in ("recompute", "hand-off"))
    fl = 2.5 * attn_flops(B, H, S, D, causal)
    ds = arms["hand-off"].fa_bwd_ds_workspace_bytes(B, H, Hkv, S, S, D) / 2 ** 30
    print(f"{name:10s} {str((B, H, Hkv, S, D)):26s} {str(dt)[6:]:9s} {str(causal):6s} {t0:13.4f} {t1:12.4f} {100 * (t0 / t1 - 1):6.1f}% "
          f"{fl / t1 / 1e9:14.1f} {ds:7.2f} {same}", flush=True)
    del q, k, v, do, o, lse, grads
    torch.cuda.empty_cache()
