"""Random shapes AND score distributions through the forward-only paths (developer tool, GPU): the fp8 kernel (head_dim 80 .. 128,
e4m3 inputs with per-tensor scales) and the wide-head kernel (head_dim 144 .. 256), against float64 on the (dequantised) inputs.
Score patterns: plain N(0,1), scaled logits, one dominant early key for every query ("sink"), one outlier key at a random place,
a per-row spread of query norms.  fp8 bound: relative Frobenius error <= 5 %, elements within 7e-2 max(1, max|ref|), LSE 1e-3.

    python tools/fuzz_fwd_paths.py [seed [cases]]      # prints FAIL lines and a final count; exit code 1 on any failure
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import flash_attention_impls_amd as fa
from oracle import attn_oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for it in range(N):
    path = str(rng.choice(["fp8", "fp8", "wide"]))
    B = int(rng.integers(1, 3)); Hkv = int(rng.integers(1, 5)); G = int(rng.choice([1, 1, 2, 4])); H = Hkv * G
    S = int(rng.choice([1, 64, 127, 128, 129, 255, 256, 257, 384, 500, 1024, 1300, 2048])) if rng.random() < 0.7 else int(rng.integers(1, 2100))
    Sk = S if rng.random() < 0.6 else max(1, S + int(rng.integers(-200, 700)))
    D = int(rng.choice([80, 96, 112, 128])) if path == "fp8" else int(rng.choice([144, 160, 192, 224, 256]))
    causal = bool(rng.random() < 0.5)
    pattern = str(rng.choice(["plain", "plain", "scaled", "sink", "outlier", "norms"]))
    if B * H * S * Sk > 3e7:
        continue
    g = torch.Generator().manual_seed(1000 + it)
    q = torch.randn(B, H, S, D, generator=g); k = torch.randn(B, Hkv, Sk, D, generator=g); v = torch.randn(B, Hkv, Sk, D, generator=g)
    if pattern == "scaled":
        q *= float(rng.choice([2.0, 3.0, 5.0]))
    elif pattern == "sink":
        u = torch.randn(D, generator=g); u *= math.sqrt(D) / u.norm()
        q += u
        k[:, :, int(rng.integers(0, min(16, Sk)))] = u * (float(rng.choice([4.0, 7.0, 10.0, 14.0])) / math.sqrt(D))
    elif pattern == "outlier":
        k[:, :, int(rng.integers(0, Sk))] = q[:, ::G, int(rng.integers(0, S))] * float(rng.choice([2.0, 4.0, 6.0]))
    elif pattern == "norms":
        q *= torch.exp(torch.randn(B, H, S, 1, generator=g) * 0.7)
    if path == "fp8":
        ds = tuple(float(t.abs().max()) / 448.0 for t in (q, k, v))
        qd, kd, vd = [(t / s).to(torch.float8_e4m3fn).cuda() for t, s in zip((q, k, v), ds)]
        o, lse = fa.flash_attn(qd, kd, vd, causal, descale=ds, return_lse=True)
        qn, kn, vn = [t.float().cpu().numpy().astype(np.float64) * s for t, s in zip((qd, kd, vd), ds)]
    else:
        qd, kd, vd = [t.to(torch.bfloat16).cuda() for t in (q, k, v)]
        o, lse = fa.flash_attn(qd, kd, vd, causal, return_lse=True)
        qn, kn, vn = [t.float().cpu().numpy() for t in (qd, kd, vd)]
    torch.cuda.synchronize()
    kn, vn = np.repeat(kn, G, axis=1), np.repeat(vn, G, axis=1)
    ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    got = o.float().cpu().numpy()
    msg = []
    if not np.isfinite(got).all():
        msg.append("nonfinite")
    emax = np.abs(got - ref).max() if ref.size else 0.0
    scale = max(1.0, np.abs(ref).max() if ref.size else 1.0)
    if path == "fp8":
        relf = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-9)
        if relf > 5e-2 or emax > 7e-2 * scale:
            msg.append(f"o relF {relf:.3f} max {emax:.3e}")
    elif emax > 1.6e-2 * scale:
        msg.append(f"o max {emax:.3e}")
    live = np.isfinite(lse_ref)
    lg = lse.cpu().numpy()
    if not np.array_equal(np.isfinite(lg), live) or (live.any() and np.abs(lg[live] - lse_ref[live]).max() > 2e-3 * max(1.0, np.abs(lse_ref[live]).max())):
        msg.append("lse")
    if it % 20 == 0:
        print("progress", it, flush=True)
    if msg:
        bad += 1
        print("FAIL", path, pattern, (B, H, Hkv, S, Sk, D, causal), msg, flush=True)
print(f"done: {bad} failures")
sys.exit(1 if bad else 0)
