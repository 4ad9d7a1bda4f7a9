"""Quick on-box check of the backward kernels against autograd through fp32 SDPA on the GPU (development aid;
the parity tests proper are tests/test_bwd_gpu.py)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from flash_attention_impls_amd import flash_attn

CASES = [
    (1, 1, 64, 128, torch.bfloat16, False),
    (1, 1, 64, 128, torch.bfloat16, True),
    (1, 2, 256, 128, torch.bfloat16, False),
    (1, 2, 256, 128, torch.bfloat16, True),
    (2, 3, 200, 128, torch.bfloat16, True),
    (1, 2, 77, 64, torch.bfloat16, False),
    (1, 2, 333, 64, torch.float16, True),
    (1, 1, 1, 128, torch.bfloat16, True),
    (2, 4, 1024, 128, torch.float16, True),
    (1, 2, 1000, 128, torch.bfloat16, False),
    (1, 9, 640, 64, torch.bfloat16, True),
]
bad = 0
for (B, H, S, D, dt, causal) in CASES:
    torch.manual_seed(S + D)
    q, k, v, do = [torch.randn(B, H, S, D, device="cuda").to(dt) for _ in range(4)]
    qf, kf, vf = [t.float().requires_grad_(True) for t in (q, k, v)]
    of = F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal, scale=1 / math.sqrt(D))
    of.backward(do.float())
    qg, kg, vg = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o = flash_attn(qg, kg, vg, causal)
    o.backward(do)
    torch.cuda.synchronize()
    tol = 1.6e-2 if dt == torch.bfloat16 else 2e-3
    line = f"{(B, H, S, D)} {str(dt)[6:]:8s} causal={causal!s:5s}"
    for name, got, ref in (("dq", qg.grad, qf.grad), ("dk", kg.grad, kf.grad), ("dv", vg.grad, vf.grad)):
        err = float((got.float() - ref).abs().max())
        lim = tol * max(1.0, float(ref.abs().max()))
        rel = float((got.float() - ref).norm() / ref.norm().clamp_min(1e-20))
        ok = err <= lim and bool(torch.isfinite(got).all())
        bad += not ok
        line += f" | {name} err {err:.2e} (lim {lim:.2e}) rel {rel:.2e} {'ok' if ok else 'FAIL'}"
    print(line, flush=True)
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
