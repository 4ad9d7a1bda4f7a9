"""GPU parity tests of grouped-query / multi-query attention: K, V with H_kv < H heads.

The reference's operator has no such case (it takes B, H, N, D from q for all three tensors, FA2-triton.py:176), so
parity is anchored on what the operator means: query head h attends to key/value head h // (H // H_kv), i.e. the result
must be the one the equal-head-count path gives on K, V expanded with ``repeat_interleave`` -- and that path is pinned
by the reference-generated goldens.  Checked three ways:
  * forward O and LSE, and dQ: BITWISE equal to the expanded run (same kernels, same arithmetic per query head);
  * dK, dV: against the float64 oracle on the expanded tensors, summed over each group (tolerances of
    test_bwd_gpu.py); they are accumulated over the group in fp32 registers, so they are at least as accurate as the
    sum of the expanded run's per-head 16-bit gradients;
  * the oracle itself on the forward (tolerances of BASELINE.md §4).
"""
import numpy as np
import pytest
import torch

from conftest import TOL, golden_gqa_names, golden_torch, load_golden
from oracle import attn_oracle as orc
from test_bwd_gpu import DT, assert_grad_close

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402


def rand_gqa(B, H, Hkv, S, D, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g).to(dtype).cuda()
    k = torch.randn(B, Hkv, S, D, generator=g).to(dtype).cuda()
    v = torch.randn(B, Hkv, S, D, generator=g).to(dtype).cuda()
    do = torch.randn(B, H, S, D, generator=g).to(dtype).cuda()
    return q, k, v, do


def expand(t, G):
    return t.repeat_interleave(G, dim=1).contiguous()


def grads(q, k, v, do, causal):
    qg, kg, vg = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    o, lse = fa.flash_attn(qg, kg, vg, causal, return_lse=True)
    o.backward(do)
    torch.cuda.synchronize()
    return o.detach(), lse.detach(), qg.grad, kg.grad, vg.grad


CASES = [
    # B, H, Hkv, S, D, dtype, causal
    (1, 4, 1, 200, 128, "bf16", True),        # multi-query
    (2, 8, 2, 256, 128, "bf16", False),
    (1, 6, 3, 129, 64, "fp16", True),
    (1, 4, 2, 513, 64, "bf16", False),
    (2, 4, 4, 96, 128, "fp16", True),         # H_kv == H through the same entry point
    (1, 8, 2, 70, 32, "bf16", True),
    (1, 12, 4, 321, 96, "bf16", True),
    (3, 2, 1, 64, 128, "bf16", False),
]


@pytest.mark.parametrize("B,H,Hkv,S,D,dt,causal", CASES)
def test_gqa_forward_backward(B, H, Hkv, S, D, dt, causal):
    G = H // Hkv
    q, k, v, do = rand_gqa(B, H, Hkv, S, D, DT[dt], seed=S + 13 * H + Hkv)
    o, lse, dq, dk, dv = grads(q, k, v, do, causal)
    assert o.shape == q.shape and lse.shape == (B, H, S)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape

    ke, ve = expand(k, G), expand(v, G)
    oe, lsee, dqe, dke, dve = grads(q, ke, ve, do, causal)
    assert torch.equal(o, oe), "O differs from the run on expanded K, V"
    assert torch.equal(lse, lsee), "LSE differs from the run on expanded K, V"
    assert torch.equal(dq, dqe), "dQ differs from the run on expanded K, V"

    # forward vs the oracle
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, ke, ve, do)]
    o_ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    err = np.abs(o.float().cpu().numpy() - o_ref).max()
    assert err <= TOL[dt] * max(1.0, np.abs(o_ref).max()), f"O: {err:.3e}"
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 2e-3

    # backward vs the float64 oracle: dK, dV of a key/value head = the sum over its query heads
    dq_ref, dk_ref, dv_ref, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=causal)
    dk_ref = dk_ref.reshape(B, Hkv, G, S, D).sum(axis=2)
    dv_ref = dv_ref.reshape(B, Hkv, G, S, D).sum(axis=2)
    tag = f"{(B, H, Hkv, S, D, dt, causal)}"
    assert_grad_close(dq, dq_ref, dt, tag + ":dq")
    assert_grad_close(dk, dk_ref, dt, tag + ":dk")
    assert_grad_close(dv, dv_ref, dt, tag + ":dv")
    # and never worse than summing the expanded run's 16-bit per-head gradients
    dk_sum = dke.float().reshape(B, Hkv, G, S, D).sum(dim=2).cpu().numpy()
    e_gqa = np.linalg.norm(dk.float().cpu().numpy() - dk_ref)
    e_sum = np.linalg.norm(dk_sum - dk_ref)
    assert e_gqa <= 1.5 * e_sum + 1e-6, (e_gqa, e_sum)


@pytest.mark.parametrize("name", golden_gqa_names())
def test_gqa_reference_generated_fixtures(name):
    """Fixtures made by the reference's own sdpa_reference on expanded K, V (oracle/gen_golden.py --gqa-only)."""
    d = load_golden(name)
    q, k, v, do = [golden_torch(d, n, "cuda") for n in ("q", "k", "v", "do")]
    o, lse, dq, dk, dv = grads(q, k, v, do, bool(d["causal"]))
    dt = d["dtype"]
    err = np.abs(o.float().cpu().numpy() - d["o"]).max()
    assert err <= TOL[dt] * max(1.0, np.abs(d["o"]).max()), err
    assert np.abs(lse.cpu().numpy() - d["lse"]).max() <= 2e-3
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert_grad_close(got, d[key], dt, f"{name}:{key}")


def test_gqa_fp8_forward_matches_expanded():
    """fp8 inputs (cfg5 is a Llama-like shape: 32 query heads, 8 key/value heads in the real model)."""
    B, H, Hkv, S = 2, 8, 2, 384
    for D, causal in ((128, True), (128, False), (64, True)):
        g = torch.Generator().manual_seed(D + causal)
        q = torch.randn(B, H, S, D, generator=g).cuda().to(torch.float8_e4m3fn)
        k = torch.randn(B, Hkv, S, D, generator=g).cuda().to(torch.float8_e4m3fn)
        v = torch.randn(B, Hkv, S, D, generator=g).cuda().to(torch.float8_e4m3fn)
        o, lse = fa.flash_attn(q, k, v, causal, return_lse=True, descale=(0.5, 2.0, 1.5))
        ke = k.view(torch.uint8).repeat_interleave(H // Hkv, dim=1).contiguous().view(torch.float8_e4m3fn)
        ve = v.view(torch.uint8).repeat_interleave(H // Hkv, dim=1).contiguous().view(torch.float8_e4m3fn)
        oe, lsee = fa.flash_attn(q, ke, ve, causal, return_lse=True, descale=(0.5, 2.0, 1.5))
        assert torch.equal(o, oe) and torch.equal(lse, lsee)
        # 5 % relative Frobenius against fp32 SDPA of the dequantised tensors (the fp8 bound of BASELINE.md §4)
        ref = torch.nn.functional.scaled_dot_product_attention(
            q.float() * 0.5, ke.float() * 2.0, ve.float() * 1.5, is_causal=causal)
        rel = (o.float() - ref).norm() / ref.norm()
        assert rel < 5e-2, rel


def test_gqa_strided_kv_views():
    """K and V as head-major views of a packed (B, S, H_kv, 2, D) buffer: strides go through unchanged."""
    B, H, Hkv, S, D = 1, 8, 2, 160, 128
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, H, S, D, generator=g).bfloat16().cuda()
    kv = torch.randn(B, S, Hkv, 2, D, generator=g).bfloat16().cuda()
    k, v = kv[:, :, :, 0].permute(0, 2, 1, 3), kv[:, :, :, 1].permute(0, 2, 1, 3)
    assert not k.is_contiguous()
    o = fa.flash_attn(q, k, v, True)
    oc = fa.flash_attn(q, k.contiguous(), v.contiguous(), True)
    assert torch.equal(o, oc)


def test_gqa_deterministic_and_large_group():
    """Eight query heads per key/value head, several key blocks: two runs are bitwise identical (no atomics)."""
    q, k, v, do = rand_gqa(1, 16, 2, 700, 128, torch.bfloat16, seed=11)
    a = grads(q, k, v, do, True)
    b = grads(q, k, v, do, True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_gqa_bad_head_counts():
    q = torch.zeros(1, 6, 32, 64, dtype=torch.bfloat16, device="cuda")
    k = torch.zeros(1, 4, 32, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(fa.FlashAttnArgumentError):
        fa.flash_attn(q, k, k)
    with pytest.raises(fa.FlashAttnArgumentError):
        fa.flash_attn(q, k[:, :3], k[:, :2])


def test_gqa_backward_split_and_unsplit_paths_agree():
    """With few key/value heads and a small batch the dK/dV kernel splits a group's query heads over several workgroups
    (fp32 partial sums + a reduction pass) when the workspace has room (fa_bwd_ex_workspace_bytes); with the smaller
    fa_bwd_workspace_bytes one workgroup streams the whole group.  Both paths against the float64 oracle and each other."""
    import ctypes
    import importlib
    host = importlib.import_module("flash_attention_impls_amd.flash_attn")
    lib = fa.load_library()
    B, H, Hkv, S, D = 1, 8, 2, 1100, 128                  # 18 query tiles x 4 heads per group: worth splitting
    q, k, v, do = rand_gqa(B, H, Hkv, S, D, torch.bfloat16, seed=5)
    scale = D ** -0.5
    o, lse = host._fwd_raw(lib, q, k, v, True, scale, None, True)
    small, big = lib.fa_bwd_workspace_bytes(B, H, S), lib.fa_bwd_ex_workspace_bytes(B, H, Hkv, S, S, D)
    assert big > small                                    # this shape is one the library wants to split
    outs = {}
    for name, nbytes in (("unsplit", small), ("split", big)):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        rc = lib.fa_bwd_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                           dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Hkv, S, S, D, *([None] * 8),
                           0, 1, scale, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.fa_last_error()
        torch.cuda.synchronize()
        outs[name] = (dq, dk, dv)
    assert torch.equal(outs["split"][0], outs["unsplit"][0])            # dQ does not depend on the split
    G = H // Hkv
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, expand(k, G), expand(v, G), do)]
    _, dk_ref, dv_ref, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=True)
    dk_ref = dk_ref.reshape(B, Hkv, G, S, D).sum(axis=2)
    dv_ref = dv_ref.reshape(B, Hkv, G, S, D).sum(axis=2)
    for name in ("unsplit", "split"):
        assert_grad_close(outs[name][1], dk_ref, "bf16", name + ":dk")
        assert_grad_close(outs[name][2], dv_ref, "bf16", name + ":dv")
    # the two differ only by fp32 summation order before the final rounding
    for a, b_ in zip(outs["split"][1:], outs["unsplit"][1:]):
        assert (a.float() - b_.float()).abs().max() <= 2 ** -7 * max(1.0, b_.float().abs().max().item())
