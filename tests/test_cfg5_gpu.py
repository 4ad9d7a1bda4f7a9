"""BASELINE config 5 at its defining size: one GPU's shard (8 of 64 batches) of (64,32,4096,128) fp8-e4m3 Q/K/V with
per-tensor scales, through flash_attn(..., descale=...) -- 256 heads, 64 K/V tiles per workgroup, the unrolled steady
state on the MX-scaled fp8 MFMAs.  The golden fixtures pin the fp8 path at S = 256 only; here the full-size launch is
checked through sampled rows recomputed in float64 on the dequantised inputs and through size-independent properties.

Stated tolerance for fp8 inputs: relative Frobenius error <= 5 %; LSE 1e-3 (relative to max(1,|lse|)).
"""
import math

import numpy as np
import pytest
import torch

from conftest import FP8_REL_FRO, FP8_ROW_MAX, FP8_TOL

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402

B, H, S, D = 8, 32, 4096, 128


def _quantise(t):
    sc = float(t.abs().max()) / 448.0
    return (t / sc).to(torch.float8_e4m3fn), sc


@pytest.fixture(scope="module")
def cfg5():
    torch.manual_seed(5)
    qs, ks, vs = [_quantise(torch.randn(B, H, S, D, device="cuda", dtype=torch.float32)) for _ in range(3)]
    (q, sq), (k, sk), (v, sv) = qs, ks, vs
    ds = (sq, sk, sv)
    o, lse = fa.flash_attn(q, k, v, False, descale=ds, return_lse=True)
    torch.cuda.synchronize()
    return q, k, v, ds, o, lse


def _row_ref(q, k, v, ds, b, h, r, n_keys):
    qq = q[b, h, r].double().cpu().numpy() * ds[0]
    kk = k[b, h, :n_keys].double().cpu().numpy() * ds[1]
    vv = v[b, h, :n_keys].double().cpu().numpy() * ds[2]
    s = kk @ qq / math.sqrt(D)
    m = s.max()
    p = np.exp(s - m)
    return (p @ vv) / p.sum(), m + math.log(p.sum())


def test_cfg5_sampled_rows_vs_f64(cfg5):
    q, k, v, ds, o, lse = cfg5
    assert o.dtype == torch.bfloat16 and o.shape == (B, H, S, D)
    rng = np.random.default_rng(5)
    picks = [(int(rng.integers(B)), int(rng.integers(H)), int(r)) for r in
             list(rng.integers(0, S, 54)) + [0, 1, 31, 32, 255, 256, 2047, 2048, 4094, 4095]]
    num = den = 0.0
    for (b, h, r) in picks:
        ref, lse_ref = _row_ref(q, k, v, ds, b, h, r, S)
        got = o[b, h, r].double().cpu().numpy()
        num += float(((got - ref) ** 2).sum())
        den += float((ref ** 2).sum())
        assert np.linalg.norm(got - ref) <= FP8_REL_FRO * np.linalg.norm(ref), (b, h, r)
        assert abs(float(lse[b, h, r]) - lse_ref) <= 1e-3 * max(1.0, abs(lse_ref))
    assert math.sqrt(num / den) <= FP8_REL_FRO
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()


def test_cfg5_determinism_and_shard_invariance(cfg5):
    """Same inputs -> bitwise the same output; a batch slice computed alone (what another rank of the 8-GPU split of
    config 5 would own) is bitwise the slice of the full launch."""
    q, k, v, ds, o, lse = cfg5
    o2, lse2 = fa.flash_attn(q, k, v, False, descale=ds, return_lse=True)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    ob = fa.flash_attn(q[3:5], k[3:5], v[3:5], False, descale=ds)
    assert torch.equal(ob, o[3:5])
    oh = fa.flash_attn(q[6:7, 9:10], k[6:7, 9:10], v[6:7, 9:10], False, descale=ds)
    assert torch.equal(oh[0, 0], o[6, 9])


def test_cfg5_value_properties(cfg5):
    """V = 1 gives O = 1 (the softmax rows sum to one); doubling the V scale doubles O exactly."""
    q, k, v, ds, o, lse = cfg5
    ones = torch.ones_like(v[:1].float()).to(torch.float8_e4m3fn)
    o1 = fa.flash_attn(q[:1], k[:1], ones, False, descale=(ds[0], ds[1], 1.0))
    assert (o1.float() - 1).abs().max() <= 2 ** -7
    o2 = fa.flash_attn(q[:2], k[:2], v[:2], False, descale=(ds[0], ds[1], 2.0 * ds[2]))
    assert torch.equal(o2, o[:2] * 2)


def test_cfg5_causal_sampled_rows(cfg5):
    """The causal variant of the shard (BASELINE.json leaves config 5's mask open: non-causal is the primary, causal the
    secondary reading) on two batches."""
    q, k, v, ds, o, lse = cfg5
    oc, lsec = fa.flash_attn(q[:2], k[:2], v[:2], True, descale=ds, return_lse=True)
    rng = np.random.default_rng(55)
    for r in list(rng.integers(0, S, 20)) + [0, 255, 256, 4095]:
        b, h = int(rng.integers(2)), int(rng.integers(H))
        ref, lse_ref = _row_ref(q, k, v, ds, b, h, int(r), int(r) + 1)
        got = oc[b, h, int(r)].double().cpu().numpy()
        assert np.linalg.norm(got - ref) <= FP8_REL_FRO * max(np.linalg.norm(ref), 1e-6), (b, h, r)
        assert abs(float(lsec[b, h, int(r)]) - lse_ref) <= 1e-3 * max(1.0, abs(lse_ref))


# ------------------------------------------------------------------ the fp8 kernel's rare paths and variants (small shapes)
def _fp8_case(B, H, Hkv, S, Sk, Dh, seed, spike=None):
    g = torch.Generator().manual_seed(seed)
    f32 = [torch.randn(B, H, S, Dh, generator=g), torch.randn(B, Hkv, Sk, Dh, generator=g), torch.randn(B, Hkv, Sk, Dh, generator=g)]
    if spike is not None:                       # key `spike[0]` aligned with query `spike[1]`: a score far above the row's first keys
        f32[1][:, :, spike[0]] = f32[0][:, : Hkv, spike[1]] * spike[2]
    ds = tuple(float(t.abs().max()) / 448.0 for t in f32)
    q, k, v = [(t / s).to(torch.float8_e4m3fn).cuda() for t, s in zip(f32, ds)]
    return q, k, v, ds


def _ref64(q, k, v, ds, causal):
    from oracle import attn_oracle as orc
    G = q.shape[1] // k.shape[1]
    deq = [t.float().cpu() * s for t, s in zip((q, k, v), ds)]
    deq[1], deq[2] = deq[1].repeat_interleave(G, 1), deq[2].repeat_interleave(G, 1)
    return orc.naive_attention_f64(*[t.numpy() for t in deq], causal=causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("key,query", [(700, 900), (130, 200), (40, 1000), (5, 1000)])
def test_fp8_forced_exact_fallback(causal, key, query):
    """A P beyond e4m3's range (here: one key whose score lies tens of binades above the reference of most rows -- outside the
    row's first block, or, (5, 1000) and (40, 1000), inside it, where it IS the reference of the rows it dominates and far
    below it for others) turns into NaN in the conversion, poisons that row's sum on the matrix pipe and sends the workgroup to the
    exact running-maximum loop: the output must come out as accurate as on ordinary data, for every row of the workgroup."""
    q, k, v, ds = _fp8_case(1, 2, 2, 1024, 1024, 128, seed=key, spike=(key, query, 6.0))
    o, lse = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
    ref, lse_ref = _ref64(q, k, v, ds, causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    of = o.double().cpu().numpy()
    assert np.linalg.norm(of - ref) <= FP8_REL_FRO * np.linalg.norm(ref)
    assert np.abs(of - ref).max() <= 7e-2 * max(1.0, np.abs(ref).max())
    assert np.abs(lse.double().cpu().numpy() - lse_ref).max() <= 1e-3 * max(1.0, np.abs(lse_ref).max())
    if not causal or key <= query:
        assert lse_ref[0, 0, query] > 30.0                   # the spike really is there


def _f64_attention(q, k, v, ds, causal=False):
    deq = [t.double().cpu().numpy() * s for t, s in zip((q, k, v), ds)]
    s = np.einsum("bhid,bhjd->bhij", deq[0], deq[1]) / math.sqrt(q.shape[-1])
    if causal:
        s = np.where(np.tril(np.ones(s.shape[-2:], dtype=bool)), s, -np.inf)
    w = np.exp(s - s.max(-1, keepdims=True))
    w /= w.sum(-1, keepdims=True)
    return np.einsum("bhij,bhjd->bhid", w, deq[2]), s


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("sink_nats", [5.0, 8.0, 10.0])
def test_fp8_attention_sink_keeps_the_small_terms(sink_nats, causal):
    """P in e4m3 spans 2^-9 .. 448 only.  With one dominant early key (an "attention sink": key 0 scores `sink_nats` above the
    rest for every query) the many small terms under it still carry a large share of the weight (1023 keys x e^-8 x e^(var/2) ~ the
    sink's own weight) and must not fall into e4m3's subnormals: the kernel places each row's window by how far its reference
    stands out of the first key tile (kGapFree8 / kPRefTop8), which is what this case exercises."""
    Bn, Hh, Sn, Dh = 1, 2, 1024, 128
    g = torch.Generator().manual_seed(int(sink_nats))
    u = torch.randn(Dh, generator=g)
    u *= math.sqrt(Dh) / u.norm()
    qf = torch.randn(Bn, Hh, Sn, Dh, generator=g) + u
    kf, vf = (torch.randn(Bn, Hh, Sn, Dh, generator=g) for _ in range(2))
    kf[:, :, 0] = u * (sink_nats / math.sqrt(Dh))                  # score of key 0: sink_nats +- sink_nats / 11 for every query
    f32 = [qf, kf, vf]
    ds = tuple(float(t.abs().max()) / 448.0 for t in f32)
    q, k, v = [(t / s).to(torch.float8_e4m3fn).cuda() for t, s in zip(f32, ds)]
    o, lse = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
    ref, s = _f64_attention(q, k, v, ds, causal)
    w0 = np.exp(s[..., 0] - np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) - s.max(-1))
    assert 0.02 < np.median(w0[..., 16:]) < 0.98                    # neither the sink nor the rest is negligible
    of = o.double().cpu().numpy()
    assert np.isfinite(of).all()
    assert np.linalg.norm(of - ref) <= FP8_REL_FRO * np.linalg.norm(ref)
    rows = np.linalg.norm(of - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
    assert np.quantile(rows, 0.99) <= 2 * FP8_REL_FRO
    lse_ref = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    assert np.abs(lse.double().cpu().numpy() - lse_ref).max() <= 1e-3 * max(1.0, np.abs(lse_ref).max())


def test_fp8_one_outlier_key_in_the_first_tile():
    """Key 5 (inside the 16 keys that fix the references) is a 6x copy of query 1000: its score against the other queries is
    N(0, 6^2) nats, so across the rows it is anything from far below to tens of binades above the rest.  Every row must stay
    within the fp8 bound (rows where key 5 dominates by more than ~15 binades lose the other keys, whose weight is then < 3 %)."""
    q, k, v, ds = _fp8_case(1, 2, 2, 1024, 1024, 128, seed=5, spike=(5, 1000, 6.0))
    o = fa.flash_attn(q, k, v, False, descale=ds).double().cpu().numpy()
    ref, _ = _f64_attention(q, k, v, ds)
    assert np.isfinite(o).all()
    assert np.linalg.norm(o - ref) <= FP8_REL_FRO * np.linalg.norm(ref)
    rows = np.linalg.norm(o - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
    assert np.quantile(rows, 0.99) <= 2 * FP8_REL_FRO and rows.max() <= FP8_ROW_MAX
    # element-wise bound on the rows the outlier does not dominate (its softmax weight under a half)
    s = _f64_attention(q, k, v, ds)[1]
    w5 = np.exp(s[..., 5] - (np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)))
    plain = w5 < 0.5
    assert plain.mean() > 0.3
    assert np.abs(o - ref)[plain].max() <= FP8_TOL * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("want_lse", [True, False])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("n_dom", [4, 12, 16, 40])
def test_fp8_many_dominant_first_keys(n_dom, causal, want_lse):
    """several comparably dominant keys at the very start (7.0 .. 7.4 nats above the rest for every query) over a
    broad tail ~10 binades below them that still carries 10 - 50 % of each row's weight at S = 4096.  A window placed from the
    first 16 keys alone sees no gap (b = 0) and rounds the whole tail to zero; the first-block sample (128 keys) contains the
    tail and opens the window downwards.  Both kernel variants (with and without the exact row sums of the LSE)."""
    Bn, Hh, Sn, Dh = 1, 2, 4096, 128
    g = torch.Generator().manual_seed(100 + n_dom)
    u = torch.randn(Dh, generator=g)
    u *= math.sqrt(Dh) / u.norm()
    qf = torch.randn(Bn, Hh, Sn, Dh, generator=g) + u
    kf, vf = (torch.randn(Bn, Hh, Sn, Dh, generator=g) for _ in range(2))
    for i in range(n_dom):
        kf[:, :, i] = u * ((7.0 + 0.4 * i / max(1, n_dom - 1)) / math.sqrt(Dh))
    ds = tuple(float(t.abs().max()) / 448.0 for t in (qf, kf, vf))
    q, k, v = [(t / s_).to(torch.float8_e4m3fn).cuda() for t, s_ in zip((qf, kf, vf), ds)]
    if want_lse:
        o, lse = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
    else:
        o = fa.flash_attn(q, k, v, causal, descale=ds)
    ref, s = _f64_attention(q, k, v, ds, causal)
    lse_ref = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    w_dom = np.exp(s[..., :n_dom] - lse_ref[..., None]).sum(-1)
    late = slice(512, None)                                          # (causal: rows that see a tail at all)
    assert 0.3 < np.median(w_dom[..., late]) < 0.95                  # neither the dominant keys nor the tail is negligible
    of = o.double().cpu().numpy()
    assert np.isfinite(of).all()
    assert np.linalg.norm(of - ref) <= FP8_REL_FRO * np.linalg.norm(ref)
    rows = np.linalg.norm(of - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
    assert np.quantile(rows, 0.99) <= 2 * FP8_REL_FRO and rows.max() <= FP8_ROW_MAX
    if want_lse:
        assert np.abs(lse.double().cpu().numpy() - lse_ref).max() <= 1e-3 * max(1.0, np.abs(lse_ref).max())


def _block0_dominant_case(Sn, Hh=1, seed=128, top=7.2, rest=5.9, n_top=2, noise=0.3):
    """Every one of a row's first 128 keys stands far above the tail: `n_top` keys at `top` nats, the other first-block keys
    at `rest`, the tail (orthogonal to the common query direction) at 0 +- `noise`.  The first-block sample then shows a maximum
    within two binades of its mean -- no gap, the e4m3 window stays high -- while every tail key lies below the window and the
    tail as a whole still carries a large share of each row's weight."""
    Dh = 128
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(Dh, generator=g)
    u *= math.sqrt(Dh) / u.norm()
    qf = noise * torch.randn(1, Hh, Sn, Dh, generator=g) + u
    kf, vf = (torch.randn(1, Hh, Sn, Dh, generator=g) for _ in range(2))
    kf = kf - (kf @ u)[..., None] * u / Dh
    kf[:, :, :128] = u * (rest / math.sqrt(Dh))
    kf[:, :, :n_top] = u * (top / math.sqrt(Dh))
    ds = tuple(float(t.abs().max()) / 448.0 for t in (qf, kf, vf))
    q, k, v = [(t / s_).to(torch.float8_e4m3fn).cuda() for t, s_ in zip((qf, kf, vf), ds)]
    return q, k, v, ds


def _rows_ref64(q, k, v, ds, h, rows, causal):
    """float64 attention of the chosen query rows of head h (batch 0) on the dequantised inputs: (O rows, weight of the keys
    behind the first 128-key block)."""
    Dh = q.shape[-1]
    qq = q[0, h, rows].double() * ds[0]
    kk = k[0, h].double() * ds[1]
    vv = v[0, h].double() * ds[2]
    s = qq @ kk.T / math.sqrt(Dh)
    if causal:
        s = s.masked_fill(torch.arange(k.shape[2], device=s.device)[None, :] > torch.as_tensor(rows, device=s.device)[:, None], float("-inf"))
    w = torch.softmax(s, dim=-1)
    return (w @ vv).cpu().numpy(), w[:, 128:].sum(-1).cpu().numpy()


@pytest.mark.parametrize("causal", [False, True])
def test_fp8_a_whole_first_block_of_dominant_keys_is_caught_by_both_variants(causal):
    """The case no first-block sample covers, at S = 4096 (the tail carries ~8 % of a late row's weight).  The variant that forms
    the exact row sums (an LSE is asked for) sees the rounded sum fall short of the exact one; the DEFAULT variant (no LSE: what
    bench.py times for config 5 and what fa_fwd_fp8(lse = NULL) runs) notices it through its sampled bound on the weight under
    the window (fa_fwd_kernel8.hpp, kUnderCap8).  Both redo the workgroup with the exact loop: the same bound as everywhere."""
    q, k, v, ds = _block0_dominant_case(4096)
    rows = np.arange(31, 4096, 32)
    ref, w_tail = _rows_ref64(q, k, v, ds, 0, rows, causal)
    assert np.median(w_tail[rows > 2048]) > 0.04                          # the tail matters ...
    o, lse = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
    o2 = fa.flash_attn(q, k, v, causal, descale=ds)                       # the default call
    o3 = fa.flash_attn(q, k, v, causal, descale=ds, fp8_checked=True)     # (exact sums without an LSE)
    assert torch.equal(o3, o)
    for got in (o, o2):
        of = got[0, 0, rows].double().cpu().numpy()
        assert np.isfinite(of).all()
        assert np.linalg.norm(of - ref) <= FP8_REL_FRO * np.linalg.norm(ref)
        rr = np.linalg.norm(of - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
        assert rr.max() <= FP8_ROW_MAX and np.quantile(rr, 0.99) <= 2 * FP8_REL_FRO


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("Sn", [32768, 65536])
def test_fp8_heavy_tail_under_the_window_at_long_sequences(Sn, causal):
    """the same input class at the long sequences this build advertises, where the tail behind a wholly
    dominant first block carries 35 - 55 % of a row's weight (S = 32768: ~24 of ~68 units; S = 65536: ~49 of ~93).  The default
    call (no LSE) must return the same accuracy as on ordinary data; sampled rows against float64."""
    q, k, v, ds = _block0_dominant_case(Sn, Hh=2, seed=Sn)
    rows = np.concatenate([np.arange(Sn // 2 + 17, Sn, Sn // 64), [Sn - 1]])
    o2 = fa.flash_attn(q, k, v, causal, descale=ds)
    for h in range(2):
        ref, w_tail = _rows_ref64(q, k, v, ds, h, rows, causal)
        assert np.median(w_tail) > (0.3 if Sn >= 65536 or not causal else 0.2), np.median(w_tail)
        of = o2[0, h, rows].double().cpu().numpy()
        assert np.isfinite(of).all()
        assert np.linalg.norm(of - ref) <= FP8_REL_FRO * np.linalg.norm(ref), (h, np.linalg.norm(of - ref) / np.linalg.norm(ref))
        rr = np.linalg.norm(of - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
        assert rr.max() <= FP8_ROW_MAX


def test_fp8_sampled_check_does_not_fire_on_ordinary_data():
    """The sampled bound must cost nothing where nothing is lost: on config 5's data the default variant returns bitwise the
    output of the variant with the exact sums (neither redoes a workgroup; a redo would show as different rounding)."""
    q, k, v, ds = _fp8_case(2, 4, 4, 4096, 4096, 128, seed=77)
    for causal in (False, True):
        o, _ = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
        assert torch.equal(fa.flash_attn(q, k, v, causal, descale=ds), o)


@pytest.mark.parametrize("shape", [(1, 2, 2, 128, 128, 128), (2, 3, 3, 333, 333, 128), (1, 4, 2, 777, 777, 128), (1, 2, 2, 1, 1, 128),
                                   (1, 2, 1, 1030, 1030, 96), (1, 2, 2, 256, 1024, 128), (1, 2, 2, 1024, 300, 80)])
@pytest.mark.parametrize("causal", [False, True])
def test_fp8_shapes_and_lse_variants(shape, causal):
    """Ragged lengths, grouped heads, other head dims (80, 96: zero-filled through the buffer bounds), own key length; the
    kernel variant that forms no exact row sums (no LSE asked for) returns bitwise the same O as the one that does."""
    B, H, Hkv, S, Sk, Dh = shape
    q, k, v, ds = _fp8_case(B, H, Hkv, S, Sk, Dh, seed=S + Dh + causal)
    o, lse = fa.flash_attn(q, k, v, causal, descale=ds, return_lse=True)
    o2 = fa.flash_attn(q, k, v, causal, descale=ds)
    assert torch.equal(o, o2)
    ref, lse_ref = _ref64(q, k, v, ds, causal)
    of = o.double().cpu().numpy()
    assert np.linalg.norm(of - ref) <= FP8_REL_FRO * max(np.linalg.norm(ref), 1e-9)
    assert np.abs(of - ref).max() <= 7e-2 * max(1.0, np.abs(ref).max())
    fin = np.isfinite(lse_ref)
    got = lse.double().cpu().numpy()
    assert np.array_equal(np.isinf(got), ~fin)               # rows without keys: O = 0, LSE = -inf
    if fin.any():
        assert np.abs(got[fin] - lse_ref[fin]).max() <= 1e-3 * max(1.0, np.abs(lse_ref[fin]).max())
    assert (of[~fin] == 0).all()


@pytest.mark.parametrize("Sn", [1024, 1280, 4096])
@pytest.mark.parametrize("want_lse", [False, True])
def test_fp8_causal_pairs_equal_single_block_launches_bitwise(Sn, want_lse):
    """The causal fp8 launch of 256 heads runs query-block PAIRS (nqb-1-t, t) with the K / V ring streaming from the long pass into
    the short one -- the short pass starts at ring stage 0 or 2, by the long pass's tile count: S = 1024 has both ((3,0): 8 tiles,
    (2,1): 6), S = 1280 also a middle block that runs alone -- and forms block 0 once where no row needs the mask.  One query block
    alone (its 256 rows against the keys up to its diagonal: S_q = 256, S_k = the block's end, bottom-right aligned mask) is a launch
    without pairs and without a streamed ring.  Both must return the same bits: a wave's arithmetic does not depend on what its
    workgroup did before."""
    torch.manual_seed(Sn)
    Bn, Hn = 8, 32
    (q, sq), (k, sk), (v, sv) = [_quantise(torch.randn(Bn, Hn, Sn, D, device="cuda", dtype=torch.float32)) for _ in range(3)]
    ds = (sq, sk, sv)
    full = fa.flash_attn(q, k, v, True, descale=ds, return_lse=want_lse)
    o_full, lse_full = full if want_lse else (full, None)
    nqb = Sn // 256
    for b, h in ((0, 0), (5, 17), (7, 31)):
        for qb in sorted({0, 1, nqb // 2, nqb - 2, nqb - 1}):
            r0, r1 = qb * 256, qb * 256 + 256
            one = fa.flash_attn(q[b:b + 1, h:h + 1, r0:r1], k[b:b + 1, h:h + 1, :r1], v[b:b + 1, h:h + 1, :r1], True, descale=ds,
                                return_lse=want_lse)
            o_one, lse_one = one if want_lse else (one, None)
            assert torch.equal(o_one.view(torch.int16), o_full[b:b + 1, h:h + 1, r0:r1].view(torch.int16)), (Sn, b, h, qb)
            if want_lse:
                assert torch.equal(lse_one, lse_full[b:b + 1, h:h + 1, r0:r1]), (Sn, b, h, qb)
    assert torch.isfinite(o_full.float()).all()
