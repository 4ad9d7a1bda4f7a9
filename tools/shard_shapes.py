"""Per-rank shard shapes of SURVEY.md 8(e)'s STRONG split, timed on ONE GPU.

For W in {1, 2, 4, 8} the shard rank 0 would own -- cfg3: 8/W batches of (., 32, 4096, 128) bf16 causal; cfg4: 16/W heads of
(1, ., 16384, 128) bf16 causal; cfg5: 64/W batches of (., 32, 4096, 128) fp8 non-causal -- is run back to back and timed.
"fraction of the W = 1 rate" is the PREDICTED per-rank efficiency of the split (compute only: the split has no data-path
collective; the final all-gather of O is `O shard MiB` over xGMI): it is NOT a scaling measurement -- no multi-GPU node was
available to this build (SCALE_r0x.json: skipped).

  python tools/shard_shapes.py > profiles/r4_shard_shapes.txt
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import flash_attention_impls_amd as fa
from flash_attention_impls_amd.bench_utils import attn_flops
from flash_attention_impls_amd.dist import shard_bounds

CFG = {  # name: (B_total, H, S, D, dtype, causal)
    "cfg3": (8, 32, 4096, 128, torch.bfloat16, True),
    "cfg4": (1, 16, 16384, 128, torch.bfloat16, True),
    "cfg5": (64, 32, 4096, 128, torch.float8_e4m3fn, False),
}


def time_units(n_units, S, D, dt, causal, iters=30):
    torch.manual_seed(0)
    descale = None
    if dt == torch.float8_e4m3fn:
        f32 = [torch.randn(1, n_units, S, D, device="cuda") for _ in range(3)]
        descale = tuple(float(t.abs().max()) / 448.0 for t in f32)
        q, k, v = [(t / s_).to(dt) for t, s_ in zip(f32, descale)]
        del f32
    else:
        q, k, v = (torch.randn(1, n_units, S, D, device="cuda").to(dt) for _ in range(3))
    for _ in range(max(3, int(0.3 / 1e-3 / max(1, n_units / 64)))):          # settle: ~0.3 s of back-to-back launches
        fa.flash_attn(q, k, v, causal, descale=descale)
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fa.flash_attn(q, k, v, causal, descale=descale)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters)
    return sorted(best)[len(best) // 2]


def main():
    lib = fa.load_library()
    print(f"# {torch.cuda.get_device_name(0)}, library {lib.fa_version()}; predicted per-rank efficiency of the strong split, NOT a scaling measurement")
    print(f"{'workload':8s} {'W':>2s} {'units/rank':>10s} {'shard (B x H, S)':>22s} {'grid':>6s} {'ms':>9s} {'TFLOP/s':>9s} {'of W=1 rate':>11s} {'O shard MiB':>11s}")
    for name, (Bt, H, S, D, dt, causal) in CFG.items():
        base = None
        for W in (1, 2, 4, 8):
            lo, hi = shard_bounds(Bt * H, 0, W)
            n = hi - lo
            ms = time_units(n, S, D, dt, causal)
            tf = attn_flops(1, n, S, D, causal) / ms / 1e9
            base = tf if base is None else base
            import ctypes
            g = ctypes.c_int(0)
            code = 2 if dt == torch.float8_e4m3fn else 0
            lib.fa_fwd_launch_info(1, n, S, D, code, int(causal), ctypes.byref(g), None, None)
            shard = f"{n // H} x {H}" if n % H == 0 and n >= H else f"1 x {n}"
            print(f"{name:8s} {W:2d} {n:10d} {shard + ', ' + str(S):>22s} {g.value:6d} {ms:9.4f} {tf:9.1f} {tf / base:11.3f} {n * S * D * 2 / 2 ** 20:11.1f}")


if __name__ == "__main__":
    main()
