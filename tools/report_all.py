"""All BASELINE.json single-GPU configurations with the reference harness' semantics
(measure_latency: 10 warm-ups, 100 per-iteration event timings, mean/std/min; code/triton_fa2/FA2-triton.py:249-268,
340-354): ms, TFLOP/s, % of the dense bf16 MFMA peak, algorithmic GB/s, tokens/s, peak memory, max|o - SDPA|;
"step MB" / "step MB rc": peak device memory (torch.cuda.max_memory_allocated, which FA2-triton.py:349 prints for its run) of one
forward + backward step with the dS hand-off the backward takes by default (a transient 2 S_q S_k bytes per query head, capped
by FA_MI355_BWD_DS_MAX_GIB) and on the recompute path (FA_MI355_BWD_DS=0); first line: what back-to-back MFMAs deliver on this
device (bench.py's `roofline.attainable`).
Last column: the same launch replayed from a captured HIP graph (100 launches per replay, per-launch ms) -- what the
kernel costs once the Python / ctypes call is out of the way; it matters for the small, launch-bound shapes."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from flash_attention_impls_amd import flash_attn
from flash_attention_impls_amd.bench_utils import attn_bytes, attn_flops, measure_latency

CONFIGS = [
    ("cfg2", 4, 8, 1024, 64, torch.bfloat16, False),
    ("cfg3", 8, 32, 4096, 128, torch.bfloat16, True),
    ("cfg3-noncausal", 8, 32, 4096, 128, torch.bfloat16, False),
    ("cfg3-fp16", 8, 32, 4096, 128, torch.float16, True),
    ("cfg4", 1, 16, 16384, 128, torch.bfloat16, True),
    ("cfg5-shard(fp8 in)", 8, 32, 4096, 128, torch.float8_e4m3fn, False),
    ("ref-main D=32", 1, 16, 1024, 32, torch.float16, True),
    ("ref-latency D=128 N=8192", 1, 32, 8192, 128, torch.float16, False),
]


def graph_latency(fn, launches=100, replays=5):
    """Per-launch ms of fn replayed from a HIP graph holding `launches` back-to-back launches (min over replays)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(launches):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(replays):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) / launches)
    return best


def step_peak_mb(q, k, v, causal):
    """Peak device memory (MB, above what is resident before) of one forward + backward step: (default backward, recompute backward)."""
    out = []
    for env in ("1", "0"):
        prev = os.environ.get("FA_MI355_BWD_DS")
        os.environ["FA_MI355_BWD_DS"] = env
        try:
            leaves = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
            d_o = torch.ones_like(q)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            flash_attn(*leaves, causal).backward(d_o)
            torch.cuda.synchronize()
            out.append((torch.cuda.max_memory_allocated() - base) / 1e6)
            del leaves, d_o
        finally:
            if prev is None:
                os.environ.pop("FA_MI355_BWD_DS", None)
            else:
                os.environ["FA_MI355_BWD_DS"] = prev
    return out


def main():
    print(torch.cuda.get_device_name(0))
    try:        # the matrix cores' measured ceiling on this device (bench.py's roofline.attainable)
        import importlib.util
        spec = importlib.util.spec_from_file_location("fa_bench_main", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        from flash_attention_impls_amd import load_library
        for name in ("bf16", "fp8"):
            att, _ = bench.attainable_mfma(load_library(), name, torch.device("cuda", 0))
            print(f"attainable ({name} MFMAs back to back on N(0,1) operands, this device): {att:.0f} TFLOP/s = {att / (5033.2 if name == 'fp8' else 2516.6):.2f} of the nominal dense peak")
    except Exception as e:  # noqa: BLE001
        print(f"attainable: not measured ({e})")
    print(f"{'config':26s} {'shape':22s} {'dtype':13s} {'causal':6s} {'mean ms':>9s} {'std':>7s} {'min ms':>8s} "
          f"{'TFLOP/s':>8s} {'%peak':>6s} {'GB/s':>7s} {'Mtok/s':>8s} {'peakMB':>8s} {'max|o-sdpa|':>11s} {'graph ms':>9s} {'step MB':>8s} {'step MB rc':>10s}", flush=True)
    for name, B, H, S, D, dt, causal in CONFIGS:
        torch.manual_seed(0)
        f32 = [torch.randn(B, H, S, D, device="cuda") for _ in range(3)]
        descale = None
        if dt == torch.float8_e4m3fn:
            descale = tuple(float(t.abs().max()) / 448.0 for t in f32)
            q, k, v = [(t / s).to(dt) for t, s in zip(f32, descale)]
            deq = [t[:1, :2].float() * s for t, s in zip((q, k, v), descale)]
        else:
            q, k, v = [t.to(dt) for t in f32]
            deq = [t[:1, :2].float() for t in (q, k, v)]
        del f32
        o = flash_attn(q, k, v, causal, descale=descale)
        # on-box SDPA (fp32 math) on a two-head slice as the accuracy spot check
        ref = F.scaled_dot_product_attention(deq[0], deq[1], deq[2], is_causal=causal, scale=1 / math.sqrt(D))
        err = float((o[:1, :2].float() - ref).abs().max())
        del deq, ref
        torch.cuda.reset_peak_memory_stats()
        m = measure_latency(lambda: flash_attn(q, k, v, causal, descale=descale), warmup=10, iters=100)
        peak_mb = torch.cuda.max_memory_allocated() / 1e6
        graph_ms = graph_latency(lambda: flash_attn(q, k, v, causal, descale=descale))
        fl = attn_flops(B, H, S, D, causal)
        by = attn_bytes(B, H, S, D, in_bytes=q.element_size())
        sec = m["mean_ms"] * 1e-3
        tf = fl / sec / 1e12
        # dense MFMA peak of the arithmetic the shape runs on: fp8 inputs at head_dim > 64 feed MX-scaled fp8 MFMAs in both
        # products (5033.2); head_dim <= 64 converts to bf16 first; 16-bit inputs 2516.6
        peak = 5033.2 if (dt == torch.float8_e4m3fn and D > 64) else 2516.6
        print(f"{name:26s} {str((B, H, S, D)):22s} {str(dt)[6:]:13s} {str(causal):6s} {m['mean_ms']:9.4f} "
              f"{m['std_ms']:7.4f} {m['min_ms']:8.4f} {tf:8.1f} {100 * tf / peak:6.1f} {by / sec / 1e9:7.0f} "
              f"{B * H * S / sec / 1e6:8.1f} {peak_mb:8.0f} {err:11.3e} {graph_ms:9.4f} " +
              ("       -          -" if dt == torch.float8_e4m3fn else "{:8.0f} {:10.0f}".format(*step_peak_mb(q, k, v, causal))), flush=True)


if __name__ == "__main__":
    main()
