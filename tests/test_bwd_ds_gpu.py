"""GPU tests of the dS hand-off backward (dK/dV kernel writes dS, dQ = scale * dS . K as a streaming GEMM:
csrc/fa_bwd_dq_gemm_kernel.hpp) against the recompute backward (FA_MI355_BWD_DS=0) and the float64 oracle.

What is asserted: dQ, dK and dV are BITWISE those of the recompute path (dK / dV: the same kernel, the stores of dS are the
only difference; dQ: the GEMM adds the same 16-bit dS . K in the same key order as the recompute kernel) on a fixed grid, on
40 randomly drawn extended shapes and on cfg3 at full size; dQ meets the backward's stated tolerances against the float64
oracle (tests/test_bwd_gpu.py); both paths are run-to-run deterministic; the workspace size alone selects the path at the C ABI.
"""
import ctypes

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import TOL
from oracle import attn_oracle as orc

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402
import importlib  # noqa: E402

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
REL_FRO = {"bf16": 6e-3, "fp16": 1e-3}


def inputs(B, H, Hkv, Sq, Sk, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, Sq, D, generator=g).to(dtype).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(dtype).cuda()
    do = torch.randn(B, H, Sq, D, generator=g).to(dtype).cuda()
    return q, k, v, do


def grads(q, k, v, do, causal, monkeypatch, ds: bool):
    """(dq, dk, dv) through the library's own workspace choice, with the hand-off allowed or not.  EVERY workspace is filled
    with 0xFF (a NaN pattern in both 16-bit types and in fp32) before the launch (flash_attn._ws_poison): a dS unit, a partial sum
    or a statistics row that is read without having been written in THIS call cannot hide behind the identical stale image the
    caching allocator hands back from the previous call."""
    monkeypatch.setenv("FA_MI355_BWD_DS", "1" if ds else "0")
    monkeypatch.setattr(fmod, "_ws_poison", True)
    lib = fa.load_library()
    scale = q.shape[-1] ** -0.5
    o, lse = fmod._fwd_raw(lib, q, k, v, causal, scale, None, True)
    dims = (q.shape[0], q.shape[1], k.shape[1], q.shape[2], k.shape[2], q.shape[3])
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, dims, q.device)
    small = lib.fa_bwd_ex_workspace_bytes(*dims)
    assert (nbytes > small) == ds, "the test means to compare the two paths"
    out = fmod._bwd_raw(lib, q, k, v, o, lse, do, causal, scale)
    torch.cuda.synchronize()
    return out


def check_against_oracle(got, ref, dt, what):
    got = got.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all(), what
    err = np.abs(got - ref).max()
    assert err <= TOL[dt] * max(1.0, np.abs(ref).max()), f"{what}: max|g-ref| = {err:.3e}"
    nrm = np.linalg.norm(ref)
    if nrm > 0:
        assert np.linalg.norm(got - ref) / nrm <= REL_FRO[dt], what


CASES = [
    # B, H, Hkv, Sq, Sk, D, dtype, causal
    (1, 1, 1, 64, 64, 128, "bf16", True), (1, 1, 1, 64, 64, 128, "bf16", False),
    (2, 2, 2, 128, 128, 64, "bf16", True), (1, 3, 3, 129, 129, 64, "fp16", False),
    (1, 2, 2, 255, 255, 128, "bf16", True), (1, 2, 2, 257, 257, 128, "fp16", True),
    (2, 1, 1, 383, 383, 128, "bf16", False), (1, 5, 5, 96, 96, 64, "bf16", True),
    (1, 1, 1, 31, 31, 128, "fp16", True), (1, 1, 1, 33, 33, 64, "bf16", False),
    (1, 2, 2, 1030, 1030, 128, "bf16", True),             # several query-block pairs, ragged end
    (1, 9, 9, 320, 320, 64, "fp16", False), (1, 1, 1, 2, 2, 64, "fp16", True),
    (1, 2, 2, 130, 130, 96, "bf16", True), (1, 2, 2, 130, 130, 48, "bf16", False), (1, 2, 2, 200, 200, 16, "fp16", True),
    (1, 8, 2, 300, 300, 128, "bf16", True), (2, 4, 1, 200, 200, 64, "fp16", False),       # grouped / multi-query heads
    (1, 2, 2, 128, 512, 128, "bf16", True), (1, 2, 2, 100, 357, 64, "bf16", True),         # longer key history (mask offset)
    (1, 2, 2, 300, 77, 128, "bf16", True), (1, 2, 2, 300, 77, 128, "bf16", False),         # fewer keys than queries
    (1, 8, 1, 1100, 1100, 128, "bf16", True),             # a group split over workgroups (fp32 partial sums) + hand-off
    (1, 1, 1, 4096, 4096, 128, "bf16", True),             # 32 key blocks for 256 CUs: the query range of a key block split 8 ways
    (1, 2, 2, 2304, 2304, 64, "fp16", False),             # the same without the mask, head_dim 64
]


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,dt,causal", CASES)
def test_handoff_against_recompute_and_oracle(B, H, Hkv, Sq, Sk, D, dt, causal, monkeypatch):
    q, k, v, do = inputs(B, H, Hkv, Sq, Sk, D, DT[dt], seed=Sq * 7 + D + (Sk - Sq))
    dq1, dk1, dv1 = grads(q, k, v, do, causal, monkeypatch, ds=True)
    dq0, dk0, dv0 = grads(q, k, v, do, causal, monkeypatch, ds=False)
    assert torch.equal(dk1.view(torch.int16), dk0.view(torch.int16)), "dK differs from the recompute path"
    assert torch.equal(dv1.view(torch.int16), dv0.view(torch.int16)), "dV differs from the recompute path"
    # dQ: the same sums of the same 16-bit dS . K in the same order (see test_handoff_bitwise_on_random_shapes)
    d = (dq1.float() - dq0.float()).abs().max().item()
    assert torch.equal(dq1.view(torch.int16), dq0.view(torch.int16)), f"dQ: hand-off vs recompute {d:.3e}"
    G = H // Hkv
    ke, ve = [t.repeat_interleave(G, dim=1) for t in (k, v)]
    ref = orc.naive_attention_bwd_f64(*[t.float().cpu().numpy() for t in (q, ke, ve, do)], causal=causal)
    check_against_oracle(dq0, ref[0], dt, "dq (recompute path: the case itself must be well enough conditioned)")
    check_against_oracle(dq1, ref[0], dt, "dq")


# randomly drawn extended shapes: G query heads per key/value head, S_k = S_q + extra (negative: fewer keys than queries), every
# head_dim the reference accepts, both types, causal or not.  dQ, dK and dV of the hand-off must be BITWISE the recompute path's:
# both form the same 16-bit dS (scores accumulated over the head_dim in the same order, -delta as the initial dP accumulator,
# the same exp2) and add dS . K over the keys in the same 32-key steps and k-slot order.
ext_shape = st.tuples(st.integers(1, 2), st.integers(1, 3), st.integers(1, 4), st.integers(1, 600), st.integers(-200, 400),
                      st.sampled_from(list(range(16, 129, 16))), st.sampled_from(["bf16", "fp16"]), st.booleans(),
                      st.integers(0, 2 ** 16))


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(ext_shape)
def test_handoff_bitwise_on_random_shapes(case):
    B, Hkv, G, Sq, extra, D, dt, causal, seed = case
    Sk = max(1, Sq + extra)
    q, k, v, do = inputs(B, Hkv * G, Hkv, Sq, Sk, D, DT[dt], seed)
    mp = pytest.MonkeyPatch()
    try:
        a = grads(q, k, v, do, causal, mp, ds=True)
        b = grads(q, k, v, do, causal, mp, ds=False)
    finally:
        mp.undo()
    for x, y, name in zip(a, b, ("dq", "dk", "dv")):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16)), (case, name, (x.float() - y.float()).abs().max().item())


def test_handoff_is_deterministic(monkeypatch):
    q, k, v, do = inputs(2, 4, 4, 777, 777, 128, torch.bfloat16, seed=11)
    a = grads(q, k, v, do, True, monkeypatch, ds=True)
    b = grads(q, k, v, do, True, monkeypatch, ds=True)
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))


def test_batch_chunks_through_a_capped_workspace(monkeypatch):
    """A batch whose dS image passes FA_MI355_BWD_DS_MAX_GIB runs in equal batch chunks through one workspace: same bits."""
    q, k, v, do = inputs(5, 2, 2, 300, 300, 128, torch.bfloat16, seed=17)
    whole = grads(q, k, v, do, True, monkeypatch, ds=True)
    lib = fa.load_library()
    per = lib.fa_bwd_ds_workspace_bytes(1, 2, 2, 300, 300, 128)
    monkeypatch.setenv("FA_MI355_BWD_DS_MAX_GIB", repr(2.5 * per / 2 ** 30))           # two batches fit, five do not
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, (5, 2, 2, 300, 300, 128), q.device)
    assert (bc, hc) == (2, 2) and nbytes == lib.fa_bwd_ds_workspace_bytes(2, 2, 2, 300, 300, 128)   # 5 = 2 + 2 + 1
    chunked = grads(q, k, v, do, True, monkeypatch, ds=True)
    for x, y in zip(whole, chunked):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))
    monkeypatch.setenv("FA_MI355_BWD_DS_MIN_BLOCKS_PER_CU", "0")                       # (head chunks of a small shape: allowed for the test)
    monkeypatch.setenv("FA_MI355_BWD_DS_MAX_GIB", repr(0.5 * per / 2 ** 30))           # not even one batch fits: recompute path
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, (5, 2, 2, 300, 300, 128), q.device)
    assert bc == 1 and hc == 1 and nbytes == lib.fa_bwd_ds_workspace_bytes(1, 1, 1, 300, 300, 128)   # one batch does not fit: head by head
    per_head = grads(q, k, v, do, True, monkeypatch, ds=True)
    for x, y in zip(whole, per_head):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))
    monkeypatch.setenv("FA_MI355_BWD_DS_MAX_GIB", repr(0.1 * per / 2 ** 30))           # not even one head fits: recompute path
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, (5, 2, 2, 300, 300, 128), q.device)
    assert (bc, hc) == (5, 2) and nbytes == lib.fa_bwd_ex_workspace_bytes(5, 2, 2, 300, 300, 128)


def test_head_group_chunks(monkeypatch):
    """Where one batch alone passes the cap, the launch is split by groups of query heads that share a key/value head: same bits."""
    dims = (2, 8, 2, 300, 300, 128)                        # 4 query heads per key/value head
    q, k, v, do = inputs(*dims, torch.bfloat16, seed=23)
    whole = grads(q, k, v, do, True, monkeypatch, ds=True)
    lib = fa.load_library()
    group = lib.fa_bwd_ds_workspace_bytes(1, 4, 1, 300, 300, 128)
    monkeypatch.setenv("FA_MI355_BWD_DS_MAX_GIB", repr(1.5 * group / 2 ** 30))         # one group fits, a batch (two groups) does not
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, dims, q.device)
    assert (bc, hc) == (2, 8) and nbytes == lib.fa_bwd_ex_workspace_bytes(*dims)       # head chunks of so small a shape would not fill the chip
    monkeypatch.setenv("FA_MI355_BWD_DS_MIN_BLOCKS_PER_CU", "0")
    bc, hc, ws, nbytes = fmod._bwd_plan(lib, dims, q.device)
    assert (bc, hc) == (1, 4) and nbytes == group
    chunked = grads(q, k, v, do, True, monkeypatch, ds=True)
    for x, y in zip(whole, chunked):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))


def test_head_image_beyond_2gib(monkeypatch):
    """S = 33024: one head's dS image is 2.03 GiB -- past what a 32-bit buffer offset reaches; both kernels address it through
    descriptors of two slab rows.  Bitwise the recompute path's gradients (the float64 oracle does not fit such a shape)."""
    q, k, v, do = inputs(1, 2, 2, 33024, 33024, 128, torch.bfloat16, seed=29)
    lib = fa.load_library()
    assert lib.fa_bwd_ds_workspace_bytes(1, 2, 2, 33024, 33024, 128) > 2 * 2 ** 31
    a = grads(q, k, v, do, True, monkeypatch, ds=True)
    b = grads(q, k, v, do, True, monkeypatch, ds=False)
    for x, y in zip(a, b):
        assert torch.isfinite(x.float()).all()
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))


def test_masked_dq_rows_are_zero(monkeypatch):
    """S_k < S_q under the causal mask: the first S_q - S_k queries see no key, their dQ rows are exactly zero."""
    q, k, v, do = inputs(1, 2, 2, 300, 77, 128, torch.bfloat16, seed=3)
    dq, _, _ = grads(q, k, v, do, True, monkeypatch, ds=True)
    assert not dq[:, :, : 300 - 77].any()
    assert dq[:, :, 300 - 77:].any()


def test_workspace_size_selects_the_path():
    """C ABI: fa_bwd_ex with fa_bwd_ds_workspace_bytes runs the hand-off, with less the recompute path; the garbage a
    too-small workspace never holds is not read (the recompute call gets a workspace poisoned with NaN patterns)."""
    lib = fa.load_library()
    B, H, S, D = 1, 4, 600, 128
    q, k, v, do = inputs(B, H, H, S, S, D, torch.bfloat16, seed=21)
    scale = D ** -0.5
    o, lse = fmod._fwd_raw(lib, q, k, v, True, scale, None, True)
    small = lib.fa_bwd_ex_workspace_bytes(B, H, H, S, S, D)
    big = lib.fa_bwd_ds_workspace_bytes(B, H, H, S, S, D)
    assert big > small + 2 * B * H * S * S
    assert lib.fa_bwd_ds_workspace_bytes(1, 1, 1, 40000, 40000, 128) > 2 ** 31      # one head's image may pass 2 GiB (per-tile descriptors)
    outs = []
    for nbytes in (big, big - 1, small):
        ws = torch.full((big,), 0xFF, dtype=torch.uint8, device="cuda")            # 0xFFFF: a NaN in both 16-bit types
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        st = fmod._strides3
        rc = lib.fa_bwd_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                           dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, H, S, S, D,
                           st(q), st(k), st(v), st(o), st(do), st(dq), st(dk), st(dv),
                           fmod._dtype_code(q.dtype), 1, scale, ws.data_ptr(), nbytes,
                           torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.fa_last_error().decode()
        torch.cuda.synchronize()
        assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all()
        outs.append((dq, dk, dv))
    assert torch.equal(outs[1][0].view(torch.int16), outs[2][0].view(torch.int16))       # both smaller sizes: the recompute path
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))       # dK the same on both paths
    ref = orc.naive_attention_bwd_f64(*[t.float().cpu().numpy() for t in (q, k, v, do)], causal=True)
    check_against_oracle(outs[0][0], ref[0], "bf16", "dq (hand-off, poisoned workspace)")


def test_full_size_cfg3_handoff(monkeypatch):
    """BASELINE cfg3 at its defining size through the hand-off (8 GiB of dS): sampled rows of dQ against the float64
    oracle of those rows, dK / dV bitwise the recompute path's."""
    B, H, S, D = 8, 32, 4096, 128
    q, k, v, do = inputs(B, H, H, S, S, D, torch.bfloat16, seed=2)
    dq1, dk1, dv1 = grads(q, k, v, do, True, monkeypatch, ds=True)
    dq0, dk0, dv0 = grads(q, k, v, do, True, monkeypatch, ds=False)
    assert torch.equal(dk1.view(torch.int16), dk0.view(torch.int16))
    assert torch.equal(dv1.view(torch.int16), dv0.view(torch.int16))
    d = (dq1.float() - dq0.float()).abs().max().item()
    assert torch.equal(dq1.view(torch.int16), dq0.view(torch.int16)), d
    for (b, h) in ((0, 0), (7, 31), (3, 17)):
        ref = orc.naive_attention_bwd_f64(*[t[b:b + 1, h:h + 1].float().cpu().numpy() for t in (q, k, v, do)], causal=True)
        check_against_oracle(dq1[b:b + 1, h:h + 1], ref[0], "bf16", f"dq[{b},{h}]")


def test_workspace_cap_bounds_the_transient_memory(monkeypatch):
    """the dS hand-off is a transient allocation on top of the step's tensors, and FA_MI355_BWD_DS_MAX_GIB bounds
    it: torch.cuda.max_memory_allocated over one forward + backward step with the default cap (the whole image: 2 S_q S_k bytes
    per query head), with a cap under the image (equal batch chunks through a smaller workspace) and with the hand-off off."""
    B, H, S, D = 4, 8, 2048, 128
    q, k, v, do = inputs(B, H, H, S, S, D, torch.bfloat16, seed=31)
    lib = fa.load_library()
    image = lib.fa_bwd_ds_workspace_bytes(B, H, H, S, S, D)
    small = lib.fa_bwd_ex_workspace_bytes(B, H, H, S, S, D)
    assert image >= 2 * B * H * S * S

    def peak(env):
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        fa.flash_attn(*leaves, True).backward(do)
        torch.cuda.synchronize()
        return torch.cuda.max_memory_allocated() - base, [t.grad for t in leaves]

    full, g_full = peak({"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": "16"})
    capped, g_cap = peak({"FA_MI355_BWD_DS": "1", "FA_MI355_BWD_DS_MAX_GIB": repr(0.3 * image / 2 ** 30)})      # one batch per chunk
    off, g_off = peak({"FA_MI355_BWD_DS": "0"})
    assert full - off >= 0.9 * (image - small), (full, off, image)                  # the image is what the hand-off adds ...
    assert capped - off <= 0.3 * image + (8 << 20), (capped, off, image)            # ... and the cap bounds it
    assert capped < full
    for a, b_, c in zip(g_full, g_cap, g_off):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16)) and torch.equal(a.view(torch.int16), c.view(torch.int16))
    info = fmod.bwd_plan_info((B, H, S, D), (B, H, S, D), q.device)                   # (environment: hand-off off)
    assert info["handoff"] is False and info["chunks"] == 1 and info["ds_bytes_full"] == image
