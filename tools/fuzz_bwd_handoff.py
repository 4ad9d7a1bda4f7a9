"""Random shapes through the backward on both paths (developer tool, GPU): the dS hand-off must return BITWISE the recompute path's
dQ, dK, dV.  Larger shapes than the hypothesis test of tests/test_bwd_ds_gpu.py: several query-block pairs, grouped heads, key
lengths on both sides of S_q, split dK/dV launches, every head_dim.

    python tools/fuzz_bwd_handoff.py [seed [cases]]     # prints FAIL lines and a final count; exit code 1 on any failure
"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
lib = fmod.load_library()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
bad = taken = 0
for it in range(N):
    B = int(rng.integers(1, 4)); Hkv = int(rng.integers(1, 9)); G = int(rng.choice([1, 1, 2, 4, 8]))
    S = int(rng.choice([1, 31, 64, 255, 256, 257, 511, 768, 1000, 1024, 1500, 2048, 3000, 4096])) if rng.random() < 0.6 else int(rng.integers(1, 3500))
    Sk = S if rng.random() < 0.5 else max(1, S + int(rng.integers(-600, 1500)))
    D = int(rng.choice([16, 32, 48, 64, 80, 96, 112, 128])); dt = str(rng.choice(["bf16", "fp16"])); causal = bool(rng.random() < 0.6)
    if B * Hkv * G * S * Sk > 3e8:
        continue
    g = torch.Generator().manual_seed(it)
    q = torch.randn(B, Hkv * G, S, D, generator=g).to(DT[dt]).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    do = torch.randn(B, Hkv * G, S, D, generator=g).to(DT[dt]).cuda()
    scale = D ** -0.5
    o, lse = fmod._fwd_raw(lib, q, k, v, causal, scale, None, True)
    os.environ["FA_MI355_BWD_DS"] = "1"
    dims = (B, Hkv * G, Hkv, S, Sk, D)
    if fmod._bwd_plan(lib, dims, q.device)[3] <= lib.fa_bwd_ex_workspace_bytes(*dims):
        continue                                     # the rule declines the hand-off for this shape: nothing to compare
    a = fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
    os.environ["FA_MI355_BWD_DS"] = "0"
    b = fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
    torch.cuda.synchronize()
    taken += 1
    for x, y, name in zip(a, b, ("dq", "dk", "dv")):
        if not torch.equal(x.view(torch.int16), y.view(torch.int16)) or not torch.isfinite(x.float()).all():
            bad += 1
            print(f"FAIL {dims} {dt} causal={causal} {name}: max diff {(x.float() - y.float()).abs().max().item():.3e}", flush=True)
    if it % 20 == 0:
        print("progress", it, flush=True)
print(f"done: {taken} shapes compared, {bad} failures")
sys.exit(1 if bad else 0)
