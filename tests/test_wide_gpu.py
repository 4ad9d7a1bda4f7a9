"""GPU parity tests of the wide-head forward and backward (head_dim 144 .. 256), csrc/fa_fwd_kernel_wide.hpp
and csrc/fa_bwd_wide_kernel.hpp.

The reference rejects these shapes (assert D <= 128, FA2-triton.py:178), so no reference-generated fixture exists for them:
parity is anchored on the float64 oracle, whose restatement of `sdpa_reference` has no head_dim-dependent branch and is
pinned by the reference's own outputs at D <= 128 (tests/test_oracle.py), and on properties that tie the wide kernel to the
pinned head_dim-128 path (zero-padded columns must not change the result).  Tolerances: those of test_parity_gpu.py.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import TOL
from oracle import attn_oracle as orc

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def rand(B, H, Hkv, Sq, Sk, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, Sq, D, generator=g).to(dtype).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    return q, k, v


def check(q, k, v, causal, dt, o, lse):
    G = q.shape[1] // k.shape[1]
    ke, ve = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    qn, kn, vn = [t.float().cpu().numpy() for t in (q, ke, ve)]
    o_ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    assert o.shape == q.shape and o.dtype == q.dtype
    err = np.abs(o.float().cpu().numpy() - o_ref).max()
    assert err <= TOL[dt] * max(1.0, np.abs(o_ref).max()), f"O: {err:.3e}"
    live = np.isfinite(lse_ref)
    got = lse.cpu().numpy()
    assert np.array_equal(np.isfinite(got), live)
    if live.any():
        assert np.abs(got[live] - lse_ref[live]).max() <= 1e-3 * max(1.0, np.abs(lse_ref[live]).max())
    if (~live).any():                                   # queries without a visible key: O = 0, LSE = -inf
        assert (o.float().cpu().numpy()[~live] == 0).all() and np.isneginf(got[~live]).all()


CASES = [
    # B, H, Hkv, Sq, Sk, D, dtype, causal
    (1, 2, 2, 128, 128, 256, "bf16", False),      # one workgroup, two tiles
    (1, 2, 2, 300, 300, 256, "bf16", True),       # ragged rows and keys, three query blocks
    (2, 4, 2, 257, 700, 192, "bf16", True),       # grouped key/value heads, a key length of its own (offset 443)
    (1, 3, 3, 64, 64, 144, "fp16", False),        # the narrowest wide head: 112 zero-filled columns
    (1, 2, 1, 1, 300, 256, "bf16", True),         # one query (decoding step)
    (1, 2, 2, 200, 77, 160, "fp16", False),       # fewer keys than queries
    (1, 2, 2, 300, 100, 256, "bf16", True),       # causal with fewer keys: the first 200 queries see no key
    (1, 2, 2, 130, 130, 208, "fp16", True),
    (1, 2, 2, 513, 515, 240, "bf16", True),       # offset 1
    (1, 2, 2, 96, 96, 224, "bf16", False),
    (1, 3, 3, 333, 1, 256, "bf16", False),        # one key: O = V
    (1, 8, 8, 1024, 1024, 256, "bf16", True),     # 64 workgroups, 16 tiles on the last block
    (2, 2, 2, 2048, 2048, 256, "fp16", False),
]


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,dt,causal", CASES)
def test_wide_head_forward_vs_f64(B, H, Hkv, Sq, Sk, D, dt, causal):
    q, k, v = rand(B, H, Hkv, Sq, Sk, D, DT[dt], seed=Sq + 7 * Sk + D)
    o, lse = fa.flash_attn(q, k, v, causal, return_lse=True)
    torch.cuda.synchronize()
    check(q, k, v, causal, dt, o, lse)
    o2, lse2 = fa.flash_attn(q, k, v, causal, return_lse=True)          # bitwise deterministic
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    assert torch.equal(fa.flash_attn(q, k, v, causal), o)                # the LSE is optional


@pytest.mark.parametrize("causal", [False, True])
def test_wide_kernel_agrees_with_the_pinned_head_dim_128_path_on_padded_inputs(causal):
    """head_dim 128 inputs padded with 16 zero columns run on the wide kernel; scores, softmax and the first 128 output
    columns are mathematically those of the head_dim-128 problem (same softmax scale given explicitly): the two kernels --
    the second one pinned by the reference's fixtures -- must agree to rounding, and the padded output columns must be zero."""
    q, k, v = rand(1, 4, 4, 384, 384, 128, torch.bfloat16, seed=3)
    scale = 1.0 / 128 ** 0.5
    o128, lse128 = fa.flash_attn(q, k, v, causal, softmax_scale=scale, return_lse=True)
    pad = lambda t: torch.cat([t, torch.zeros(*t.shape[:-1], 16, dtype=t.dtype, device=t.device)], dim=-1)
    o144, lse144 = fa.flash_attn(pad(q), pad(k), pad(v), causal, softmax_scale=scale, return_lse=True)
    assert (o144[..., 128:] == 0).all()
    assert (o144[..., :128].float() - o128.float()).abs().max() <= TOL["bf16"]
    assert (lse144 - lse128).abs().max() <= 1e-3


def test_wide_head_strided_inputs_fp32_round_trip_and_large_logits():
    B, H, S, D = 2, 3, 200, 256
    g = torch.Generator().manual_seed(9)
    # (B, S, H, D) storage viewed as (B, H, S, D): the kernel addresses it in place (no host copy)
    qs, ks, vs = [torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).cuda().transpose(1, 2) for _ in range(3)]
    o, lse = fa.flash_attn(qs, ks, vs, True, return_lse=True)
    check(qs.contiguous(), ks.contiguous(), vs.contiguous(), True, "bf16", o, lse)
    # fp32 inputs are computed in fp16 and cast back (FA2-triton.py:241-244)
    qf, kf, vf = [torch.randn(1, 2, 150, 192, generator=g).cuda() for _ in range(3)]
    of = fa.flash_attn(qf, kf, vf, False)
    assert of.dtype == torch.float32
    ref, _ = orc.naive_attention_f64(*[t.half().float().cpu().numpy() for t in (qf, kf, vf)], causal=False)
    assert np.abs(of.cpu().numpy() - ref).max() <= TOL["fp32"] * max(1.0, np.abs(ref).max())
    # logits of +-60: the running maximum keeps every exponential <= 1
    q, k, v = rand(1, 2, 2, 256, 256, 256, torch.bfloat16, seed=4)
    o, lse = fa.flash_attn(q * 6, k * 6, v, True, return_lse=True)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    check(q * 6, k * 6, v, True, "bf16", o, lse)


BWD_CASES = [
    # B, H, Hkv, Sq, Sk, D, dtype, causal
    (1, 2, 2, 128, 128, 256, "bf16", False),
    (1, 2, 2, 300, 300, 256, "bf16", True),       # ragged rows and keys, three query blocks
    (2, 4, 2, 257, 700, 192, "bf16", True),       # grouped key/value heads (dK, dV summed over the group), offset 443
    (1, 3, 3, 64, 64, 144, "fp16", False),        # the narrowest wide head
    (1, 2, 1, 1, 300, 256, "bf16", True),         # one query
    (1, 2, 2, 200, 77, 160, "fp16", False),       # fewer keys than queries, image rows padded 77 -> 80
    (1, 2, 2, 300, 100, 256, "bf16", True),       # causal with fewer keys: the first 200 queries see no key (their dQ = 0)
    (1, 6, 2, 513, 515, 240, "bf16", True),       # groups of three, offset 2
    (2, 2, 2, 1024, 1024, 256, "bf16", True),     # eight query blocks per head, longest first
]


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,dt,causal", BWD_CASES)
def test_wide_head_backward_vs_f64(B, H, Hkv, Sq, Sk, D, dt, causal):
    """dQ, dK, dV of head_dim 144 .. 256 (csrc/fa_bwd_wide_kernel.hpp writes P and dS, three library GEMMs finish) against the
    float64 oracle; parity unpinned by the reference, which stops at head_dim 128 -- anchored like the forward (module docstring)."""
    from test_bwd_gpu import assert_grad_close
    q, k, v = rand(B, H, Hkv, Sq, Sk, D, DT[dt], seed=Sq * 7 + Sk)
    do = torch.randn(B, H, Sq, D, generator=torch.Generator().manual_seed(D)).to(DT[dt]).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o = fa.flash_attn(*leaves, causal)
    o.backward(do)
    dq, dk, dv = [t.grad for t in leaves]
    assert dk.shape == k.shape and dv.shape == v.shape and dq.shape == q.shape
    G = H // Hkv
    ke, ve = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, ke, ve, do)]
    dq_ref, dk_ref, dv_ref, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=causal)
    dk_ref = dk_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    dv_ref = dv_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    assert_grad_close(dq, dq_ref, dt, "dq")
    assert_grad_close(dk, dk_ref, dt, "dk")
    assert_grad_close(dv, dv_ref, dt, "dv")


def test_wide_head_backward_properties():
    """Zero-padded columns do not change the gradients of the columns that exist (head_dim 160 inside 256 against the same data as
    head_dim 160), strided inputs, chunked images (a cap of a few MiB: one key/value group per launch) and determinism."""
    import importlib
    import os
    fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
    g = torch.Generator().manual_seed(11)
    B, H, Hkv, S, D = 1, 4, 2, 384, 160
    q, do = [torch.randn(B, H, S, D, generator=g).bfloat16().cuda() for _ in range(2)]
    k, v = [torch.randn(B, Hkv, S, D, generator=g).bfloat16().cuda() for _ in range(2)]

    def grads(qq, kk, vv, dd, causal=True):
        leaves = [t.clone().requires_grad_(True) for t in (qq, kk, vv)]
        fa.flash_attn(*leaves, causal, softmax_scale=D ** -0.5).backward(dd)
        return [t.grad for t in leaves]

    base = grads(q, k, v, do)
    again = grads(q, k, v, do)
    assert all(torch.equal(a, b_) for a, b_ in zip(base, again))
    # the same problem embedded in head_dim 256 (zero columns): identical gradients on the first 160 columns, zeros beyond
    pad = lambda t: torch.nn.functional.pad(t, (0, 256 - D))          # noqa: E731
    wide = grads(pad(q), pad(k), pad(v), pad(do))
    for a, b_, name in zip(base, wide, ("dq", "dk", "dv")):
        assert torch.equal(a, b_[..., :D]), name
        assert (b_[..., D:] == 0).all(), name
    # strided views (a (B, S, H, D) buffer seen as (B, H, S, D))
    qs = q.transpose(1, 2).contiguous().transpose(1, 2)
    ks = k.transpose(1, 2).contiguous().transpose(1, 2)
    strided = grads(qs, ks, v, do)
    assert all(torch.equal(a, b_) for a, b_ in zip(base, strided))
    # one key/value group per launch
    prev = os.environ.get("FA_MI355_BWD_WIDE_MAX_GIB")
    os.environ["FA_MI355_BWD_WIDE_MAX_GIB"] = "0.001"
    try:
        chunked = grads(q, k, v, do)
    finally:
        if prev is None:
            os.environ.pop("FA_MI355_BWD_WIDE_MAX_GIB", None)
        else:
            os.environ["FA_MI355_BWD_WIDE_MAX_GIB"] = prev
    assert all(torch.equal(a, b_) for a, b_ in zip(base, chunked))
    assert fmod.MAX_HEAD_DIM == 256


def test_wide_head_rules_backward_fp8_and_launch_info():
    q, k, v = rand(1, 2, 2, 64, 64, 256, torch.bfloat16, seed=1)
    o = fa.flash_attn(q.clone().requires_grad_(True), k, v, False)                                # differentiable since 0.1.36
    assert o.requires_grad
    with pytest.raises(fa.FlashAttnArgumentError):
        fa.flash_attn(*rand(1, 1, 1, 32, 32, 272, torch.bfloat16, seed=1), False)              # > 256
    q8 = torch.zeros(1, 1, 32, 256, device="cuda").to(torch.float8_e4m3fn)
    with pytest.raises(fa.FlashAttnArgumentError):
        fa.flash_attn(q8, q8, q8, False, descale=(1.0, 1.0, 1.0))                                # fp8 stops at 128
    lib = fa.load_library()
    assert lib.fa_supported(0, 256) == 1 and lib.fa_supported(1, 144) == 1 and lib.fa_supported(2, 144) == 0
    grid, block, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.fa_fwd_launch_info(2, 8, 1000, 256, 0, 1, ctypes.byref(grid), ctypes.byref(block), ctypes.byref(lds)) == 0
    assert (grid.value, block.value, lds.value) == (2 * 8 * 8, 512, 131072)           # 128-row workgroups of 8 waves x 16 rows
