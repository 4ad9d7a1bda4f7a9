"""GPU parity tests of a key/value length of its own (S_k != S_q), forward and backward.

Not a reference case (its q, k, v share one N, FA2-triton.py:176), so parity is anchored on the part that is pinned:
with the bottom-right aligned causal mask, S_q queries against S_k keys are the LAST S_q queries of the square S_k
problem -- the product's own square path (pinned by the reference-generated goldens) and the float64 oracle (whose
extension is checked against its square case in tests/test_oracle_bwd.py) must both agree with it.
Tolerances: those of test_parity_gpu.py / test_bwd_gpu.py.
"""
import numpy as np
import pytest
import torch

from conftest import TOL, golden_kvlen_names, golden_torch, load_golden
from oracle import attn_oracle as orc
from test_bwd_gpu import DT, assert_grad_close

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402


def rand(B, H, Hkv, Sq, Sk, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, Sq, D, generator=g).to(dtype).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    do = torch.randn(B, H, Sq, D, generator=g).to(dtype).cuda()
    return q, k, v, do


def grads(q, k, v, do, causal):
    qg, kg, vg = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    o, lse = fa.flash_attn(qg, kg, vg, causal, return_lse=True)
    o.backward(do)
    torch.cuda.synchronize()
    return o.detach(), lse.detach(), qg.grad, kg.grad, vg.grad


CASES = [
    # B, H, Hkv, Sq, Sk, D, dtype, causal
    (1, 2, 2, 64, 256, 128, "bf16", True),        # offset a multiple of every tile size
    (1, 2, 2, 100, 229, 128, "bf16", True),       # ragged, offset 129: the diagonal crosses 32-row blocks
    (2, 3, 3, 1, 300, 128, "bf16", True),         # one query (decoding step): sees every key
    (1, 2, 2, 17, 48, 64, "fp16", True),
    (1, 4, 2, 260, 700, 128, "bf16", True),       # several query blocks, grouped key/value heads
    (1, 2, 2, 300, 301, 64, "bf16", True),        # offset 1
    (1, 2, 1, 130, 1000, 96, "bf16", True),
    (1, 2, 2, 200, 77, 128, "bf16", False),       # fewer keys than queries: non-causal only
    (2, 2, 2, 64, 640, 64, "fp16", False),
    (1, 3, 3, 333, 1, 128, "bf16", False),        # one key: O = V, LSE = score
    (1, 2, 2, 513, 515, 32, "bf16", True),
    (1, 2, 2, 300, 100, 128, "bf16", True),       # fewer keys than queries, causal: the first 200 queries see no key
    (2, 4, 2, 129, 64, 64, "fp16", True),
    (1, 1, 1, 700, 3, 128, "bf16", True),
]


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,dt,causal", CASES)
def test_other_key_length_forward_backward(B, H, Hkv, Sq, Sk, D, dt, causal):
    G = H // Hkv
    q, k, v, do = rand(B, H, Hkv, Sq, Sk, D, DT[dt], seed=Sq * 3 + Sk)
    o, lse, dq, dk, dv = grads(q, k, v, do, causal)
    assert o.shape == q.shape and lse.shape == (B, H, Sq) and dq.shape == q.shape
    assert dk.shape == k.shape and dv.shape == v.shape
    for t in (o, dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    if causal and Sk < Sq:                        # queries without a visible key: O = 0, LSE = -inf, no gradient
        n = Sq - Sk
        assert (o[:, :, :n] == 0).all() and torch.isneginf(lse[:, :, :n]).all() and (dq[:, :, :n] == 0).all()
        assert torch.isfinite(lse[:, :, n:]).all()
    else:
        assert torch.isfinite(lse).all()

    ke, ve = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, ke, ve, do)]
    o_ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    err = np.abs(o.float().cpu().numpy() - o_ref).max()
    assert err <= TOL[dt] * max(1.0, np.abs(o_ref).max()), f"O: {err:.3e}"
    live = np.isfinite(lse_ref)
    assert np.array_equal(np.isfinite(lse.cpu().numpy()), live)
    assert np.abs(lse.cpu().numpy()[live] - lse_ref[live]).max() <= 2e-3
    dq_ref, dk_ref, dv_ref, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=causal)
    dk_ref = dk_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    dv_ref = dv_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    tag = f"{(B, H, Hkv, Sq, Sk, D, dt, causal)}"
    assert_grad_close(dq, dq_ref, dt, tag + ":dq")
    assert_grad_close(dk, dk_ref, dt, tag + ":dk")
    assert_grad_close(dv, dv_ref, dt, tag + ":dv")


@pytest.mark.parametrize("name", golden_kvlen_names())
def test_reference_generated_key_length_fixtures(name):
    """Fixtures from the reference's sdpa_reference on the square S_k problem (its last S_q rows): the one way the
    reference itself can speak about S_q != S_k (oracle/gen_golden.py --ext-only)."""
    d = load_golden(name)
    q, k, v, do = [golden_torch(d, n, "cuda") for n in ("q", "k", "v", "do")]
    o, lse, dq, dk, dv = grads(q, k, v, do, True)
    dt = d["dtype"]
    err = np.abs(o.float().cpu().numpy() - d["o"]).max()
    assert err <= TOL[dt] * max(1.0, np.abs(d["o"]).max()), err
    assert np.abs(lse.cpu().numpy() - d["lse"]).max() <= 2e-3
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert_grad_close(got, d[key], dt, f"{name}:{key}")


@pytest.mark.parametrize("Sq,Sk", [(2048, 2304), (2048, 2560), (1792, 2112)])
def test_paired_causal_blocks_with_a_longer_key_sequence(Sq, Sk):
    """A launch large enough for the paired schedule (two query blocks per workgroup, the staging ring running on from the first
    block into the second -- `cont` in fa_fwd_kernel16.hpp -- whenever S_k - S_q is a multiple of 256; (1792, 2112) is the case
    where it is not and the second block gets its own prologue) with the bottom-right aligned mask: the forward of the S_q x S_k
    problem is the tail of the product's own square S_k x S_k run, block-aligned offsets bitwise."""
    B, H, D = 1, 64, 128
    g = torch.Generator().manual_seed(Sq + Sk)
    qf, k, v = [torch.randn(B, H, Sk, D, generator=g).bfloat16().cuda() for _ in range(3)]
    q = qf[:, :, Sk - Sq:].contiguous()
    o, lse = fa.flash_attn(q, k, v, True, return_lse=True)
    o_sq, lse_sq = fa.flash_attn(qf, k, v, True, return_lse=True)
    assert torch.isfinite(o.float()).all()
    assert (o.float() - o_sq[:, :, Sk - Sq:].float()).abs().max() <= TOL["bf16"]
    assert (lse - lse_sq[:, :, Sk - Sq:]).abs().max() <= 2e-3
    if (Sk - Sq) % 256 == 0:
        assert torch.equal(o, o_sq[:, :, Sk - Sq:])
    assert torch.equal(fa.flash_attn(q, k, v, True), o)


def test_paired_causal_blocks_with_a_shorter_key_sequence():
    """S_q > S_k under the bottom-right aligned mask at the paired schedule's scale: the first S_q - S_k queries see no key
    (O = 0, LSE = -inf), the others are the square S_k x S_k problem of the remaining queries -- including the workgroup whose
    second query block has no key tile at all while the ring has already staged "its" first tiles."""
    B, H, D, Sq, Sk = 1, 64, 128, 2304, 2048
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, H, Sq, D, generator=g).bfloat16().cuda()
    k, v = [torch.randn(B, H, Sk, D, generator=g).bfloat16().cuda() for _ in range(2)]
    o, lse = fa.flash_attn(q, k, v, True, return_lse=True)
    o_sq, lse_sq = fa.flash_attn(q[:, :, Sq - Sk:].contiguous(), k, v, True, return_lse=True)
    assert (o[:, :, :Sq - Sk] == 0).all() and torch.isinf(lse[:, :, :Sq - Sk]).all()
    assert torch.equal(o[:, :, Sq - Sk:], o_sq) and torch.equal(lse[:, :, Sq - Sk:], lse_sq)


@pytest.mark.parametrize("Sq,Sk,D", [(128, 512, 128), (90, 347, 128), (256, 1024, 64)])
def test_causal_tail_of_the_square_problem(Sq, Sk, D):
    """The S_q x S_k causal problem against the product's own square S_k x S_k run (rows S_k - S_q ..)."""
    B, H = 1, 2
    g = torch.Generator().manual_seed(Sq + Sk)
    qf, k, v, dof = [torch.randn(B, H, Sk, D, generator=g).bfloat16().cuda() for _ in range(4)]
    dof[:, :, :Sk - Sq] = 0                       # only the tail rows carry an upstream gradient
    q, do = qf[:, :, Sk - Sq:].contiguous(), dof[:, :, Sk - Sq:].contiguous()
    o, lse, dq, dk, dv = grads(q, k, v, do, True)
    o_sq, lse_sq, dq_sq, dk_sq, dv_sq = grads(qf, k, v, dof, True)
    tol = TOL["bf16"]
    assert (o.float() - o_sq[:, :, Sk - Sq:].float()).abs().max() <= tol
    assert (lse - lse_sq[:, :, Sk - Sq:]).abs().max() <= 2e-3
    for a, b_, name in ((dq, dq_sq[:, :, Sk - Sq:], "dq"), (dk, dk_sq, "dk"), (dv, dv_sq, "dv")):
        ref = b_.float()
        assert (a.float() - ref).abs().max() <= tol * max(1.0, ref.abs().max().item()), name
        assert ((a.float() - ref).norm() / ref.norm()).item() <= 8e-3, name


def test_other_key_length_fp8_and_strided():
    """fp8 inputs (native Q.K^T path at head_dim 128, converted path at 64) and K, V as strided views."""
    B, H, Sq, Sk = 1, 4, 150, 421
    for D in (128, 64):
        g = torch.Generator().manual_seed(D)
        q = torch.randn(B, H, Sq, D, generator=g).cuda().to(torch.float8_e4m3fn)
        k = torch.randn(B, H, Sk, D, generator=g).cuda().to(torch.float8_e4m3fn)
        v = torch.randn(B, H, Sk, D, generator=g).cuda().to(torch.float8_e4m3fn)
        for causal in (True, False):
            o = fa.flash_attn(q, k, v, causal)
            qn, kn, vn = [t.float().cpu().numpy() for t in (q, k, v)]
            ref, _ = orc.naive_attention_f64(qn, kn, vn, causal=causal)
            rel = np.linalg.norm(o.float().cpu().numpy() - ref) / np.linalg.norm(ref)
            assert rel < 5e-2, (D, causal, rel)
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, 2, 70, 128, generator=g).bfloat16().cuda()
    kv = torch.randn(1, 333, 2, 2, 128, generator=g).bfloat16().cuda()
    k, v = kv[:, :, :, 0].permute(0, 2, 1, 3), kv[:, :, :, 1].permute(0, 2, 1, 3)
    assert torch.equal(fa.flash_attn(q, k, v, True), fa.flash_attn(q, k.contiguous(), v.contiguous(), True))


def test_other_key_length_errors():
    q = torch.zeros(1, 2, 64, 64, dtype=torch.bfloat16, device="cuda")
    k = torch.zeros(1, 2, 32, 64, dtype=torch.bfloat16, device="cuda")
    fa.flash_attn(q, k, k, causal=True)                          # fewer keys than queries: allowed (rows without keys give 0)
    fa.flash_attn(q, k, k, causal=False)
    with pytest.raises(fa.FlashAttnArgumentError):
        fa.flash_attn(q, k[:, :, :0], k[:, :, :0])               # no keys


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("D", [128, 64])
def test_exact_fallback_with_grouped_heads_and_other_key_length(causal, D):
    """A score spike beyond the fixed softmax reference's range sends the workgroup through the exact online-softmax
    fallback; here with grouped key/value heads and S_k != S_q (the fallback has its own tile ranges and mask)."""
    B, H, Hkv, Sq, Sk = 1, 4, 2, 384, 900
    q, k, v, _ = rand(B, H, Hkv, Sq, Sk, D, torch.bfloat16, seed=21 + D)
    q = q * 0.25
    k[0, 0, 700] = (q[0, 1, 300] * 160.0).to(torch.bfloat16)      # key/value head 0 serves query heads 0, 1
    k[0, 1, 133] = (q[0, 2, 20] * 160.0).to(torch.bfloat16)
    o, lse = fa.flash_attn(q, k, v, causal, return_lse=True)
    G = H // Hkv
    qn, kn, vn = [t.float().cpu().numpy() for t in (q, k.repeat_interleave(G, 1), v.repeat_interleave(G, 1))]
    ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert np.abs(o.float().cpu().numpy() - ref).max() <= TOL["bf16"] * max(1.0, np.abs(ref).max())
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 2e-3 * max(1.0, np.abs(lse_ref).max())
    s = float(q[0, 1, 300].float() @ k[0, 0, 700].float()) / np.sqrt(D) * 1.4427
    assert s > 70                                                  # really beyond the fast path's range
