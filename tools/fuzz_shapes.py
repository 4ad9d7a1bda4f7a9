"""Random shapes through forward + backward against the float64 oracle (developer tool, GPU): batch, key/value heads, group
size, S_q, S_k, head_dim, dtype and mask drawn at random -- odd head counts (virtual-head grid mapping), grouped heads and
split dK/dV launches, key lengths on both sides of S_q.

    python tools/fuzz_shapes.py [seed [cases]]         # prints FAIL lines and a final count; exit code 1 on any failure
"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import flash_attention_impls_amd as fa
from oracle import attn_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TOL = {"bf16": 1.6e-2, "fp16": 2e-3}
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for it in range(N):
    B = int(rng.integers(1, 4)); Hkv = int(rng.integers(1, 8)); G = int(rng.choice([1, 1, 2, 3, 4, 8]))
    H = Hkv * G
    S = int(rng.choice([1, 17, 64, 100, 255, 256, 257, 500, 700, 1024, 1100, 1500])) if rng.random() < 0.7 else int(rng.integers(1, 1600))
    Sk = S if rng.random() < 0.5 else max(1, S + int(rng.integers(-300, 900)))
    D = int(rng.choice([16, 32, 64, 96, 128, 128, 144, 176, 208, 256])); dt = str(rng.choice(["bf16", "fp16"])); causal = bool(rng.random() < 0.6)
    if B * H * S * Sk > 2e7:      # keep the float64 oracle quick
        continue
    g = torch.Generator().manual_seed(it)
    q = torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    do = torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o, lse = fa.flash_attn(*leaves, causal, return_lse=True)
    o.backward(do); torch.cuda.synchronize()
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, k.repeat_interleave(G, 1), v.repeat_interleave(G, 1), do)]
    ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    dq_r, dk_r, dv_r, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=causal)
    dk_r = dk_r.reshape(B, Hkv, G, Sk, D).sum(2); dv_r = dv_r.reshape(B, Hkv, G, Sk, D).sum(2)
    errs = {"o": (o.detach().float().cpu().numpy(), ref), "dq": (leaves[0].grad.float().cpu().numpy(), dq_r),
            "dk": (leaves[1].grad.float().cpu().numpy(), dk_r), "dv": (leaves[2].grad.float().cpu().numpy(), dv_r)}
    msg = []
    for name, (got, r) in errs.items():
        e = np.abs(got - r).max() if r.size else 0.0
        if not np.isfinite(got).all() or e > TOL[dt] * max(1.0, np.abs(r).max() if r.size else 1.0):
            msg.append(f"{name}:{e:.3e}")
    live = np.isfinite(lse_ref)
    lg = lse.detach().cpu().numpy()
    if not np.array_equal(np.isfinite(lg), live) or (live.any() and np.abs(lg[live] - lse_ref[live]).max() > 2e-3 * max(1.0, np.abs(lse_ref[live]).max())):
        msg.append("lse")
    if it % 20 == 0:
        print("progress", it, flush=True)
    if msg:
        bad += 1
        print("FAIL", (B, H, Hkv, S, Sk, D, dt, causal), msg, flush=True)
print(f"done: {bad} failures")
sys.exit(1 if bad else 0)
