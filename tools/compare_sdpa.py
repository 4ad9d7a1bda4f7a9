"""Outside reference point on the same GPU: the attention kernel this image's own PyTorch ships (torch.nn.functional.
scaled_dot_product_attention on its flash backend -- AOTriton's tuned kernels on ROCm) timed beside this library, interleaved in
one process, forward and forward + backward, on the BASELINE shapes.  Not a parity tool (tests/ hold that) and not the reference
(whose Triton kernel cannot travel to the GPU box and does not cover head_dim 128): it answers "what does the vendor's own path
reach on this box at this board power".

  python tools/compare_sdpa.py [--rounds 5] [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

from flash_attention_impls_amd import flash_attn
from flash_attention_impls_amd.bench_utils import attn_flops

SHAPES = [
    ("cfg2", 4, 8, 1024, 64, torch.bfloat16, False),
    ("cfg3", 8, 32, 4096, 128, torch.bfloat16, True),
    ("cfg3-noncausal", 8, 32, 4096, 128, torch.bfloat16, False),
    ("cfg3-fp16", 8, 32, 4096, 128, torch.float16, True),
    ("cfg4", 1, 16, 16384, 128, torch.bfloat16, True),
    ("d64", 8, 32, 4096, 64, torch.bfloat16, True),
]


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), "torch", torch.__version__)
    print(f"{'shape':16s} {'(B,H,S,D)':22s} {'dtype':9s} {'causal':6s} | {'this fwd ms':>11s} {'TF/s':>7s} | {'torch SDPA fwd ms':>17s} {'TF/s':>7s} {'backend':>9s} | "
          f"{'ratio':>5s} | {'this f+b ms':>11s} {'TF/s':>7s} | {'torch f+b ms':>12s} {'TF/s':>7s} | {'ratio':>5s} | max|o - o_torch|", flush=True)
    for name, B, H, S, D, dt, causal in SHAPES:
        torch.manual_seed(0)
        q, k, v = [torch.randn(B, H, S, D, device="cuda").to(dt) for _ in range(3)]
        backend = None
        for cand, label in ((SDPBackend.FLASH_ATTENTION, "flash"), (SDPBackend.EFFICIENT_ATTENTION, "efficient")):
            try:
                with sdpa_kernel(cand):
                    o_t = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
                torch.cuda.synchronize()
                backend = (cand, label)
                break
            except RuntimeError:
                continue
        if backend is None:
            print(f"{name:16s} torch SDPA: no fused backend takes this shape")
            continue
        o = flash_attn(q, k, v, causal)
        err = float((o.float() - o_t.float()).abs().max())

        def ours_f():
            return flash_attn(q, k, v, causal)

        def torch_f():
            with sdpa_kernel(backend[0]):
                return F.scaled_dot_product_attention(q, k, v, is_causal=causal)

        leaves = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
        d_o = torch.randn_like(q)

        def ours_fb():
            for t in leaves:
                t.grad = None
            flash_attn(*leaves, causal).backward(d_o)

        def torch_fb():
            for t in leaves:
                t.grad = None
            with sdpa_kernel(backend[0]):
                F.scaled_dot_product_attention(*leaves, is_causal=causal).backward(d_o)

        best = {"of": 1e9, "tf": 1e9, "ob": 1e9, "tb": 1e9}
        for _ in range(a.rounds):           # interleaved: both libraries see the same thermal state
            best["of"] = min(best["of"], timed(ours_f, a.iters))
            best["tf"] = min(best["tf"], timed(torch_f, a.iters))
        for _ in range(max(1, a.rounds // 2)):
            best["ob"] = min(best["ob"], timed(ours_fb, max(3, a.iters // 4)))
            best["tb"] = min(best["tb"], timed(torch_fb, max(3, a.iters // 4)))
        fl = attn_flops(B, H, S, D, causal)
        tf = lambda ms, mult=1.0: mult * fl / (ms * 1e-3) / 1e12      # noqa: E731
        print(f"{name:16s} {str((B, H, S, D)):22s} {str(dt)[6:]:9s} {str(causal):6s} | {best['of']:11.4f} {tf(best['of']):7.1f} | {best['tf']:17.4f} {tf(best['tf']):7.1f} {backend[1]:>9s} | "
              f"{best['tf'] / best['of']:5.2f} | {best['ob']:11.4f} {tf(best['ob'], 3.5):7.1f} | {best['tb']:12.4f} {tf(best['tb'], 3.5):7.1f} | {best['tb'] / best['ob']:5.2f} | {err:.3e}", flush=True)


if __name__ == "__main__":
    main()
