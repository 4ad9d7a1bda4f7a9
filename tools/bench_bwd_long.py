"""Backward at long sequences (B = 1): recompute, the dS hand-off in head chunks under the default cap, and under a 40 GiB cap;
gradients of every arm compared bitwise with the recompute arm's (developer tool; profiles/r3_bwd_long.txt)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops
lib = fmod.load_library()
for (B, H, S, D) in ((1, 16, 32768, 128), (1, 8, 65536, 128)):
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(4))
    scale = D ** -0.5
    o, lse = fmod._fwd_raw(lib, q, k, v, True, scale, None, True)
    res = {}
    for name, env in (("recompute", {"FA_MI355_BWD_DS": "0"}), ("hand-off, 16 GiB cap (head chunks)", {"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": "16"}),
                      ("hand-off, 40 GiB cap", {"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": "40"})):
        os.environ.update(env)
        bc, hc, ws, nbytes = fmod._bwd_plan(lib, (B, H, H, S, S, D), q.device)
        del ws                      # (only the plan: a second multi-GiB block held beside the call's own is a fresh hipMalloc, and
        plan = (bc, hc, None, nbytes)   # freshly allocated VRAM can stall its first users for 0.1 - 1 s while the driver clears it)
        for _ in range(3):
            g = fmod._bwd_raw(lib, q, k, v, o, lse, do, True, scale)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g = fmod._bwd_raw(lib, q, k, v, o, lse, do, True, scale)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 4
        res[name] = g
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(g, res["recompute"]))
        print(f"{(B,H,S,D)} {name:36s} chunks (batches, heads) = {plan[0], plan[1]}  workspace {plan[3] / 2**30:6.2f} GiB  {t:8.3f} ms  "
              f"{2.5 * attn_flops(B, H, S, D, True) / t / 1e9:7.1f} TF/s  bitwise = recompute: {same}", flush=True)
        del plan
    del q, k, v, do, o, lse, res, g
    torch.cuda.empty_cache()
