"""GPU: randomly drawn shapes (hypothesis) through the whole path, forward and backward, against the float64 oracle.
Complements the fixed grids of test_parity_gpu.py / test_bwd_gpu.py: ragged lengths around every tile boundary, every
head_dim the reference accepts, both dtypes, causal or not, batch*heads not a multiple of the 8 XCD groups."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from conftest import TOL
from oracle import attn_oracle as orc

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}

shape = st.tuples(st.integers(1, 2), st.integers(1, 5), st.integers(1, 330), st.sampled_from(list(range(16, 129, 16))),
                  st.sampled_from(["bf16", "fp16"]), st.booleans(), st.integers(0, 2 ** 16))


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(shape)
def test_forward_random_shapes(case):
    B, H, S, D, dt, causal, seed = case
    g = torch.Generator().manual_seed(seed)
    q, k, v = [torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda() for _ in range(3)]
    o, lse = fa.flash_attn(q, k, v, causal, return_lse=True)
    ref, lse_ref = orc.naive_attention_f64(*[t.float().cpu().numpy() for t in (q, k, v)], causal=causal)
    got = o.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), case
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 1e-3 * max(1.0, np.abs(lse_ref).max()), case


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(shape)
def test_backward_random_shapes(case):
    B, H, S, D, dt, causal, seed = case
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = [torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda() for _ in range(4)]
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    fa.flash_attn(*leaves, causal).backward(do)
    torch.cuda.synchronize()
    ref = orc.naive_attention_bwd_f64(*[t.float().cpu().numpy() for t in (q, k, v, do)], causal=causal)
    for leaf, r, key in zip(leaves, ref[:3], ("dq", "dk", "dv")):
        got = leaf.grad.float().cpu().numpy()
        assert np.isfinite(got).all(), (case, key)
        assert np.abs(got - r).max() <= TOL[dt] * max(1.0, np.abs(r).max()), (case, key)


# grouped key/value heads and a key/value length of its own (the extended entry points): G query heads per key/value
# head, S_k = S + extra keys (extra < 0: fewer keys than queries; under the bottom-right aligned causal mask the first
# queries then see no key: O = 0, LSE = -inf)
ext_shape = st.tuples(st.integers(1, 2), st.integers(1, 3), st.integers(1, 4), st.integers(1, 300), st.integers(-150, 260),
                      st.sampled_from(list(range(16, 129, 16))), st.sampled_from(["bf16", "fp16"]), st.booleans(),
                      st.integers(0, 2 ** 16))


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(ext_shape)
def test_forward_backward_random_extended_shapes(case):
    B, Hkv, G, S, extra, D, dt, causal, seed = case
    Sk = max(1, S + extra)
    H = Hkv * G
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda()
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(DT[dt]).cuda()
    do = torch.randn(B, H, S, D, generator=g).to(DT[dt]).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o, lse = fa.flash_attn(*leaves, causal, return_lse=True)
    o.backward(do)
    torch.cuda.synchronize()
    ke, ve = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    qn, kn, vn, don = [t.float().cpu().numpy() for t in (q, ke, ve, do)]
    ref, lse_ref = orc.naive_attention_f64(qn, kn, vn, causal=causal)
    got = o.detach().float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), case
    live = np.isfinite(lse_ref)
    lse_got = lse.detach().cpu().numpy()
    assert np.array_equal(np.isfinite(lse_got), live), case
    if live.any():
        assert np.abs(lse_got[live] - lse_ref[live]).max() <= 1e-3 * max(1.0, np.abs(lse_ref[live]).max()), case
    dq_ref, dk_ref, dv_ref, _ = orc.naive_attention_bwd_f64(qn, kn, vn, don, causal=causal)
    dk_ref = dk_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    dv_ref = dv_ref.reshape(B, Hkv, G, Sk, D).sum(axis=2)
    for leaf, r, key in zip(leaves, (dq_ref, dk_ref, dv_ref), ("dq", "dk", "dv")):
        gotg = leaf.grad.float().cpu().numpy()
        assert gotg.shape == r.shape and np.isfinite(gotg).all(), (case, key)
        assert np.abs(gotg - r).max() <= TOL[dt] * max(1.0, np.abs(r).max()), (case, key)


# the two forward-only paths of round 2: fp8 inputs on the fp8 kernel (head_dim 80 .. 128) with score patterns that move
# each row's e4m3 window (a dominant early key, an outlier key anywhere, scaled logits), and the wide-head kernel (144 .. 256)
fwd_only = st.tuples(st.sampled_from(["fp8", "wide"]), st.integers(1, 2), st.integers(1, 3), st.sampled_from([1, 2, 4]),
                     st.integers(1, 700), st.integers(-200, 400), st.integers(0, 4), st.booleans(),
                     st.sampled_from(["plain", "scaled", "sink", "outlier"]), st.integers(0, 2 ** 16))


@settings(max_examples=50, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(fwd_only)
def test_forward_random_shapes_fp8_and_wide_heads(case):
    import math
    from conftest import FP8_REL_FRO, FP8_TOL
    path, B, Hkv, G, S, extra, dsel, causal, pattern, seed = case
    Sk = max(1, S + extra)
    H = Hkv * G
    D = [80, 96, 112, 128, 128][dsel] if path == "fp8" else [144, 160, 192, 224, 256][dsel]
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g)
    k = torch.randn(B, Hkv, Sk, D, generator=g)
    v = torch.randn(B, Hkv, Sk, D, generator=g)
    if pattern == "scaled":
        q = q * 3.0
    elif pattern == "sink":                              # key 0 (or a later one of the first tile) scores ~ 8 nats for every query
        u = torch.randn(D, generator=g)
        u = u * (math.sqrt(D) / u.norm())
        q = q + u
        k[:, :, seed % min(16, Sk)] = u * (8.0 / math.sqrt(D))
    elif pattern == "outlier":
        k[:, :, seed % Sk] = q[:, ::G, seed % S] * 4.0
    if path == "fp8":
        ds = tuple(float(t.abs().max()) / 448.0 for t in (q, k, v))
        qd, kd, vd = [(t / s_).to(torch.float8_e4m3fn).cuda() for t, s_ in zip((q, k, v), ds)]
        o, lse = fa.flash_attn(qd, kd, vd, causal, descale=ds, return_lse=True)
        qn, kn, vn = [t.float().cpu().numpy().astype(np.float64) * s_ for t, s_ in zip((qd, kd, vd), ds)]
    else:
        qd, kd, vd = [t.to(torch.bfloat16).cuda() for t in (q, k, v)]
        o, lse = fa.flash_attn(qd, kd, vd, causal, return_lse=True)
        qn, kn, vn = [t.float().cpu().numpy() for t in (qd, kd, vd)]
    ref, lse_ref = orc.naive_attention_f64(qn, np.repeat(kn, G, axis=1), np.repeat(vn, G, axis=1), causal=causal)
    got = o.float().cpu().numpy()
    assert np.isfinite(got).all(), case
    scale = max(1.0, np.abs(ref).max())
    if path == "fp8":
        assert np.linalg.norm(got - ref) <= FP8_REL_FRO * max(np.linalg.norm(ref), 1e-9), case
        assert np.abs(got - ref).max() <= FP8_TOL * scale, case
    else:
        assert np.abs(got - ref).max() <= TOL["bf16"] * scale, case
    live = np.isfinite(lse_ref)
    lg = lse.cpu().numpy()
    assert np.array_equal(np.isfinite(lg), live), case
    if live.any():
        assert np.abs(lg[live] - lse_ref[live]).max() <= 2e-3 * max(1.0, np.abs(lse_ref[live]).max()), case
