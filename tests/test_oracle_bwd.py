"""CPU: pin the backward oracle (numpy + plain-C restatements of FA2-triton.py:98-170) against the
backward fixtures oracle/gen_golden.py produced by autograd through the REFERENCE's sdpa_reference and,
where it can be interpreted, from the reference's own Triton _bwd_kernel."""
import numpy as np
import pytest
import torch

from conftest import c_oracle_bwd, golden_bwd_names, golden_f32, golden_torch, load_golden
from oracle import attn_oracle as orc

NAMES = golden_bwd_names()


def test_bwd_golden_set_present():
    assert len(NAMES) >= 8
    for must in ("bwd_fp16_d64_causal", "bwd_bf16_d128_causal", "bwd_bf16_d128_ragged", "bwd_bf16_d128_s1"):
        assert must in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_naive_bwd_f64_matches_reference_autograd(name):
    d = load_golden(name)
    q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
    dq, dk, dv, delta = orc.naive_attention_bwd_f64(q, k, v, do, causal=bool(d["causal"]))
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        ref = d[key]
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), key
    # delta stored in the fixture uses the dtype-rounded forward output: equal up to that rounding
    tol = {"fp16": 4e-3, "bf16": 3e-2}[d["dtype"]]
    assert np.abs(delta - d["delta"]).max() <= tol * max(1.0, np.abs(delta).max())


@pytest.mark.parametrize("name", NAMES)
def test_sdpa_bwd_oracle_matches_reference_autograd(name):
    d = load_golden(name)
    q, k, v, do = [golden_torch(d, n) for n in ("q", "k", "v", "do")]
    o, dq, dk, dv = orc.sdpa_bwd_oracle(q, k, v, do, causal=bool(d["causal"]))
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert torch.allclose(got, torch.from_numpy(d[key]), atol=2e-6, rtol=1e-5), key
    # the fixture's o is the same fp32 result rounded to the storage dtype (half an ulp)
    ulp = {"fp16": 2.0 ** -11, "bf16": 2.0 ** -8}[d["dtype"]]
    assert torch.allclose(o, golden_torch(d, "o").float(), atol=1e-6, rtol=1.01 * ulp)


@pytest.mark.parametrize("name", NAMES)
def test_c_bwd_oracle_matches_reference_autograd(name, oracle_clib):
    d = load_golden(name)
    q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
    outs = c_oracle_bwd(oracle_clib, q, k, v, do, bool(d["causal"]))
    for got, key in zip(outs, ("dq", "dk", "dv")):
        ref = d[key]
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), key


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("blocks", [(128, 128), (64, 32)])
def test_tiled_recompute_bwd(name, blocks):
    """Tile loop of the reference kernel (corrected dS), statistics from the fixture's lse/delta."""
    d = load_golden(name)
    if d["S"] > 320 and blocks == (64, 32):
        pytest.skip("slow in pure numpy")
    q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
    dq, dk, dv = orc.tiled_recompute_bwd(q, k, v, do, d["lse"], causal=bool(d["causal"]),
                                         block_m=blocks[0], block_n=blocks[1])
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        ref = d[key]
        assert np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max()), key


@pytest.mark.parametrize("name", [n for n in NAMES if "dv_kernel" in load_golden(n)])
def test_reference_kernel_dv_agrees_dq_dk_defective(name):
    """The interpreted reference _bwd_kernel: dV equals autograd of sdpa_reference (fp16 atomics);
    dQ/dK do not (FA2-triton.py:160-161) -- the defect is recorded, not reproduced."""
    d = load_golden(name)
    assert np.abs(d["dv_kernel"].astype(np.float32) - d["dv"]).max() < 4e-3
    assert np.abs(d["dq_kernel"].astype(np.float32) - d["dq"]).max() > 1.0
    assert np.abs(d["dk_kernel"].astype(np.float32) - d["dk"]).max() > 1.0


def test_mfma_rounding_model_within_tolerance():
    """p and dS rounded to bf16 before the gradient products (what the HIP kernels do) stays well inside
    the backward tolerance used by the GPU tests."""
    d = load_golden("bwd_bf16_d128_causal")
    q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
    got = orc.tiled_recompute_bwd(q, k, v, do, d["lse"], causal=True, w_dtype="bf16", delta=d["delta"])
    for g, key in zip(got, ("dq", "dk", "dv")):
        ref = d[key]
        rel = np.linalg.norm(g - ref) / np.linalg.norm(ref)
        assert rel < 6e-3, (key, rel)


def test_oracle_other_key_length_is_the_tail_of_the_square_problem():
    """N_k != N (not a reference case): with the bottom-right aligned causal mask, the N queries against N_k keys are
    exactly the last N queries of the square N_k problem -- which ties this extension of the oracle to the part the
    reference's fixtures pin.  Gradients: dq of those rows; dk, dv when only those rows carry an upstream gradient."""
    rng = np.random.default_rng(7)
    B, H, N, Nk, D = 1, 2, 37, 90, 32
    qf, k, v, dof = [rng.standard_normal((B, H, Nk, D)) for _ in range(4)]
    q, do = qf[:, :, Nk - N:], dof[:, :, Nk - N:]
    o, lse = orc.naive_attention_f64(q, k, v, causal=True)
    o_sq, lse_sq = orc.naive_attention_f64(qf, k, v, causal=True)
    assert np.array_equal(o, o_sq[:, :, Nk - N:]) and np.array_equal(lse, lse_sq[:, :, Nk - N:])
    do_masked = dof.copy()
    do_masked[:, :, :Nk - N] = 0.0
    dq, dk, dv, _ = orc.naive_attention_bwd_f64(q, k, v, do, causal=True)
    dq_sq, dk_sq, dv_sq, _ = orc.naive_attention_bwd_f64(qf, k, v, do_masked, causal=True)
    assert np.allclose(dq, dq_sq[:, :, Nk - N:], rtol=0, atol=1e-13)
    assert np.allclose(dk, dk_sq, rtol=0, atol=1e-13) and np.allclose(dv, dv_sq, rtol=0, atol=1e-13)
    # non-causal: plain cross attention, every key visible
    o_nc, _ = orc.naive_attention_f64(q, k, v, causal=False)
    s = np.einsum("bhid,bhjd->bhij", q, k) / np.sqrt(D)
    p = np.exp(s - s.max(-1, keepdims=True))
    assert np.allclose(o_nc, np.einsum("bhij,bhjd->bhid", p / p.sum(-1, keepdims=True), v), atol=1e-14)


def test_oracle_queries_without_keys_give_zero():
    """N_k < N under the bottom-right aligned causal mask: the first N - N_k queries see no key -> O = 0, LSE = -inf,
    dq = 0 (the l == 0 guard of flash_attn_cutlass.cu:446-452); the remaining queries are the square N_k problem."""
    rng = np.random.default_rng(3)
    B, H, N, Nk, D = 1, 1, 50, 20, 16
    q, do = rng.standard_normal((B, H, N, D)), rng.standard_normal((B, H, N, D))
    k, v = rng.standard_normal((B, H, Nk, D)), rng.standard_normal((B, H, Nk, D))
    o, lse = orc.naive_attention_f64(q, k, v, causal=True)
    assert (o[:, :, :N - Nk] == 0).all() and np.isneginf(lse[:, :, :N - Nk]).all()
    o_sq, lse_sq = orc.naive_attention_f64(q[:, :, N - Nk:], k, v, causal=True)
    assert np.array_equal(o[:, :, N - Nk:], o_sq) and np.array_equal(lse[:, :, N - Nk:], lse_sq)
    dq, dk, dv, _ = orc.naive_attention_bwd_f64(q, k, v, do, causal=True)
    dq_sq, dk_sq, dv_sq, _ = orc.naive_attention_bwd_f64(q[:, :, N - Nk:], k, v, do[:, :, N - Nk:], causal=True)
    assert (dq[:, :, :N - Nk] == 0).all() and np.allclose(dq[:, :, N - Nk:], dq_sq, atol=1e-14)
    assert np.allclose(dk, dk_sq, atol=1e-13) and np.allclose(dv, dv_sq, atol=1e-13)


def test_oracle_matches_reference_generated_grouped_head_fixtures():
    """gqa_*.npz: the reference's own sdpa_reference on K, V expanded with repeat_interleave (oracle/gen_golden.py).
    Pins what "query head h attends to key/value head h // G" means, forward and backward, for the oracle."""
    from conftest import golden_f32, golden_gqa_names, load_golden
    names = golden_gqa_names()
    assert len(names) == 2
    for name in names:
        d = load_golden(name)
        G = d["H"] // int(d["Hkv"])
        q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
        ke, ve = np.repeat(k, G, axis=1), np.repeat(v, G, axis=1)
        o, lse = orc.naive_attention_f64(q, ke, ve, causal=bool(d["causal"]))
        assert np.abs(o - d["o"]).max() < 2e-5, name
        assert np.abs(lse - d["lse"]).max() < 1e-4, name
        dq, dk, dv, _ = orc.naive_attention_bwd_f64(q, ke, ve, do, causal=bool(d["causal"]))
        B, Hkv, S, D = k.shape
        dk, dv = dk.reshape(B, Hkv, G, S, D).sum(2), dv.reshape(B, Hkv, G, S, D).sum(2)
        for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
            assert np.abs(got - d[key]).max() <= 2e-4 * max(1.0, np.abs(d[key]).max()), (name, key)


def test_oracle_matches_reference_generated_key_length_fixtures():
    """kvlen_*.npz: the reference's sdpa_reference on the square S_k problem, last S_q query rows stored
    (oracle/gen_golden.py): pins the bottom-right aligned causal mask of the oracle, forward and backward."""
    from conftest import golden_f32, golden_kvlen_names, load_golden
    names = golden_kvlen_names()
    assert len(names) == 2
    for name in names:
        d = load_golden(name)
        q, k, v, do = [golden_f32(d, n) for n in ("q", "k", "v", "do")]
        assert q.shape[2] == d["S"] and k.shape[2] == int(d["Sk"]) != d["S"]
        o, lse = orc.naive_attention_f64(q, k, v, causal=True)
        assert np.abs(o - d["o"]).max() < 2e-5 and np.abs(lse - d["lse"]).max() < 1e-4, name
        dq, dk, dv, _ = orc.naive_attention_bwd_f64(q, k, v, do, causal=True)
        for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
            assert np.abs(got - d[key]).max() <= 2e-4 * max(1.0, np.abs(d[key]).max()), (name, key)
