"""Probe for the sporadic long stall seen in tools/bench_bwd_long.py with multi-GiB hand-off workspaces: every launch set timed on
its own (events + synchronize), three rounds over the arms, allocator statistics before and after."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
lib = fmod.load_library()
B, H, S, D = 1, 8, 65536, 128
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(4))
scale = D ** -0.5
o, lse = fmod._fwd_raw(lib, q, k, v, True, scale, None, True)
arms = (("recompute", {"FA_MI355_BWD_DS": "0"}), ("hand-off 40 GiB cap (2 chunks of 4 heads)", {"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": "40"}),
        ("hand-off 80 GiB cap (one launch set)", {"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": "80"}))
for rnd in range(3):
    for name, env in arms:
        os.environ.update(env)
        st0 = torch.cuda.memory_stats()
        ts, hs = [], []
        for i in range(10):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            g = fmod._bwd_raw(lib, q, k, v, o, lse, do, True, scale)
            e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)); hs.append((t1 - t0) * 1e3)
        st1 = torch.cuda.memory_stats()
        print(f"round {rnd} {name:44s} GPU ms per launch set: " + " ".join(f"{t:7.1f}" for t in ts) +
              f" | host ms max {max(hs):7.1f} | hipMalloc calls {st1['num_device_alloc'] - st0['num_device_alloc']} hipFree {st1['num_device_free'] - st0['num_device_free']}"
              f" reserved {st1['reserved_bytes.all.current'] / 2**30:.1f} GiB", flush=True)
