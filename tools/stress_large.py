"""One-off large-shape checks (developer tool; the test suite keeps to shapes that finish in seconds):
very long sequences, very many heads, row strides close to the 2 GiB-per-head addressing limit.
Forward against sampled rows of a float64 reference, backward against PyTorch's own SDPA autograd on the GPU."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_attention_impls_amd as fa  # noqa: E402


def rows_f64(q, k, v, rows, causal):
    """float64 attention for a few query rows of head (0, 0)."""
    qd, kd, vd = q[0, 0, rows].double(), k[0, 0].double(), v[0, 0].double()
    s = qd @ kd.T / math.sqrt(q.shape[-1])
    if causal:
        off = k.shape[2] - q.shape[2]
        mask = torch.arange(k.shape[2], device=q.device)[None, :] > (rows[:, None] + off)
        s = s.masked_fill(mask, float("-inf"))
    return torch.softmax(s, -1) @ vd, torch.logsumexp(s, -1)


def check(name, B, H, S, D, dtype, causal, bwd=True, Sk=None, Hkv=None):
    torch.manual_seed(1)
    Sk, Hkv = Sk or S, Hkv or H
    q = torch.randn(B, H, S, D, device="cuda").to(dtype).requires_grad_(bwd)
    k = torch.randn(B, Hkv, Sk, D, device="cuda").to(dtype).requires_grad_(bwd)
    v = torch.randn(B, Hkv, Sk, D, device="cuda").to(dtype).requires_grad_(bwd)
    o, lse = fa.flash_attn(q, k, v, causal, return_lse=True)
    rows = torch.tensor(sorted({0, 1, 31, 32, S // 3, S // 2 + 5, S - 2, S - 1} & set(range(S))), device="cuda")
    ref, lse_ref = rows_f64(q.detach(), k.detach(), v.detach(), rows, causal)
    e_o = (o[0, 0, rows].double() - ref).abs().max().item()
    e_l = (lse[0, 0, rows].double() - lse_ref).abs().max().item()
    msg = f"{name:34s} ({B},{H}/{Hkv},{S}x{Sk},{D}) causal={int(causal)}: max|o-ref| {e_o:.2e}  max|lse-ref| {e_l:.2e}"
    ok = e_o < 1.6e-2 and e_l < 2e-3
    if bwd:
        do = torch.randn_like(o)
        o.backward(do)
        q2, k2, v2 = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
        G = H // Hkv
        ke, ve = (k2.repeat_interleave(G, 1), v2.repeat_interleave(G, 1)) if G > 1 else (k2, v2)
        if Sk == S:
            o2 = F.scaled_dot_product_attention(q2, ke, ve, is_causal=causal)
        else:
            off = Sk - S
            mask = None
            if causal:
                mask = torch.arange(Sk, device="cuda")[None, :] <= (torch.arange(S, device="cuda")[:, None] + off)
            o2 = F.scaled_dot_product_attention(q2, ke, ve, attn_mask=mask)
        o2.backward(do)
        for a, b_, nm in ((q.grad, q2.grad, "dq"), (k.grad, k2.grad, "dk"), (v.grad, v2.grad, "dv")):
            rel = ((a.float() - b_.float()).norm() / b_.float().norm()).item()
            msg += f"  {nm} rel {rel:.2e}"
            ok = ok and rel < 2e-2
    print(msg + ("  OK" if ok else "  FAIL"), flush=True)
    return ok


def main():
    ok = True
    ok &= check("long sequence", 1, 1, 65536, 128, torch.bfloat16, True)
    ok &= check("long sequence, non-causal, D=64", 1, 2, 32768, 64, torch.float16, False)
    ok &= check("many heads", 96, 40, 128, 64, torch.bfloat16, True)
    ok &= check("many heads, grouped", 64, 64, 256, 128, torch.bfloat16, True, Hkv=8)
    ok &= check("long keys, few queries", 2, 8, 128, 128, torch.bfloat16, True, Sk=131072, bwd=False)
    ok &= check("chunk of a long prefill", 1, 8, 2048, 128, torch.bfloat16, True, Sk=32768, Hkv=2)
    # row stride close to the addressing limit: S * stride * 2 bytes just under 2 GiB per (batch, head) slice
    S, D, pad = 4096, 128, 2 ** 17 - 4096
    big = torch.randn(1, 1, S, pad + D, device="cuda").to(torch.bfloat16)
    q = big[..., :D]
    o = fa.flash_attn(q, q, q, True)
    o2 = fa.flash_attn(q.contiguous(), q.contiguous(), q.contiguous(), True)
    same = torch.equal(o, o2)
    print(f"row stride {pad + D} elements ({(S + 256) * (pad + D) * 2 / 2 ** 30:.2f} GiB span): strided == contiguous: {same}")
    ok &= same
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
