"""CPU: pin the oracle (numpy restatement + plain-C restatement) against the golden vectors
that oracle/gen_golden.py produced from the REFERENCE's own sdpa_reference / _fwd_kernel."""
import math

import numpy as np
import pytest
import torch

from conftest import c_oracle_fwd, golden_f32, golden_names, golden_rows, golden_small_names, golden_torch, load_golden
from oracle import attn_oracle as orc

NAMES = golden_names()


def test_golden_set_present():
    assert len(NAMES) >= 10
    for must in ("cfg1_fp32_nc", "cfg1_fp32_causal", "bf16_d128_causal", "bf16_d128_ragged", "fp16_d64_x8"):
        assert must in NAMES


def test_regeneration_anchors():
    """Anchors recorded in SURVEY.md §8c for the seeded inputs/outputs."""
    d = load_golden("cfg1_fp32_nc")
    assert np.allclose(d["q"][0, 0, 0, :3], [-1.12584, -1.15236, -0.25058], atol=1e-5)
    assert abs(float(d["o"].sum()) - (-85.466233)) < 1e-3
    assert abs(float(np.abs(d["o"]).sum()) - 978.561608) < 1e-2
    d = load_golden("cfg1_fp32_causal")
    assert abs(float(d["o"].sum()) - (-209.128734)) < 1e-3


def _inputs(d):
    if d["dtype"] == "fp8":
        ds = d["descale"]
        return [golden_f32(d, n) * ds[i] for i, n in enumerate("qkv")]
    return [golden_f32(d, n) for n in "qkv"]


def _ref_o(d):
    return d["o"].astype(np.float32) if d["dtype"] == "fp8" else d["o_f32"]


@pytest.mark.parametrize("name", NAMES)
def test_sdpa_oracle_matches_reference(name):
    """sdpa_oracle == reference sdpa_reference output (bit-for-bit in the storage dtype)."""
    d = load_golden(name)
    if d["dtype"] == "fp8":
        q, k, v = [torch.from_numpy(x) for x in _inputs(d)]
        o = orc.sdpa_oracle(q, k, v, causal=bool(d["causal"]))
        assert torch.equal(golden_rows(d, o), torch.from_numpy(d["o"]))
        return
    q, k, v = [golden_torch(d, n) for n in "qkv"]
    o = orc.sdpa_oracle(q, k, v, causal=bool(d["causal"]))
    assert torch.equal(o, golden_torch(d, "o"))


@pytest.mark.parametrize("name", NAMES)
def test_naive_f64_matches_reference(name):
    d = load_golden(name)
    q, k, v = _inputs(d)
    o, lse = orc.naive_attention_f64(q, k, v, causal=bool(d["causal"]))
    o, lse = golden_rows(d, o), golden_rows(d, lse)
    assert np.abs(o - _ref_o(d)).max() < 2e-5 * max(1.0, np.abs(_ref_o(d)).max())
    assert np.abs(lse - d["lse"]).max() < 1e-4


@pytest.mark.parametrize("name", golden_small_names())
@pytest.mark.parametrize("deferred", [False, True])
def test_tiled_restatement_matches_reference(name, deferred):
    """_fwd_kernel restatement (reference arithmetic and the build's deferred arithmetic)."""
    d = load_golden(name)
    q, k, v = _inputs(d)
    o, m, l = orc.tiled_online_softmax(q, k, v, causal=bool(d["causal"]), block_m=128, block_n=128,
                                       deferred=deferred, skip_masked_tiles=deferred)
    ref = _ref_o(d)
    assert np.abs(o - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(m + np.log(l) - d["lse"]).max() < 1e-4


@pytest.mark.parametrize("name", [n for n in NAMES if "m" in load_golden(n)])
def test_tiled_restatement_matches_reference_kernel_stats(name):
    """m, l stored by the reference's own Triton _fwd_kernel (interpreted, D<=64)."""
    d = load_golden(name)
    q, k, v = _inputs(d)
    o, m, l = orc.tiled_online_softmax(q, k, v, causal=bool(d["causal"]), block_m=128, block_n=128)
    assert np.abs(m - d["m"]).max() < 1e-3 * max(1.0, np.abs(d["m"]).max())
    assert np.abs(l - d["l"]).max() < 2e-3 * np.abs(d["l"]).max()
    # the reference kernel itself (fp16 output) against its own sdpa_reference
    assert np.abs(d["o_kernel"].astype(np.float32) - o).max() < 2e-2 * max(1.0, np.abs(o).max())


@pytest.mark.parametrize("name", golden_small_names())
def test_c_oracle_matches_reference(name, oracle_clib):
    d = load_golden(name)
    q, k, v = _inputs(d)
    o, lse = c_oracle_fwd(oracle_clib, q, k, v, bool(d["causal"]), block_m=64, block_n=64)
    ref = _ref_o(d)
    assert np.abs(o - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(lse - d["lse"]).max() < 1e-4


def test_c_oracle_tile_size_independent(oracle_clib):
    d = load_golden("bf16_d128_ragged")
    q, k, v = _inputs(d)
    o1, l1 = c_oracle_fwd(oracle_clib, q, k, v, True, block_m=256, block_n=64)
    o2, l2 = c_oracle_fwd(oracle_clib, q, k, v, True, block_m=32, block_n=17)
    assert np.abs(o1 - o2).max() < 1e-5
    assert np.abs(l1 - l2).max() < 1e-5


def test_p_rounding_model_within_tolerance():
    """The MFMA path rounds P to bf16 before PV: the restatement with that rounding stays inside
    the stated bf16 tolerance -> the tolerance is achievable by construction."""
    d = load_golden("bf16_d128_causal")
    q, k, v = _inputs(d)
    o, _, _ = orc.tiled_online_softmax(q, k, v, causal=True, block_m=256, block_n=64, deferred=True,
                                       skip_masked_tiles=True, p_dtype="bf16")
    ref = d["o_f32"]
    assert np.abs(orc.round_to_dtype(o, "bf16") - ref).max() < 1.6e-2 * max(1.0, np.abs(ref).max())


def test_sym_rel_err_metric(oracle_clib):
    a = np.array([1.0, -2.0, 0.0, 1e-6], np.float32)
    b = np.array([1.1, -2.0, 0.0, 2e-6], np.float32)
    e = orc.sym_rel_err(a, b)
    assert abs(e - max(0.1 / (2.1 + 1e-5), 1e-6 / (3e-6 + 1e-5))) < 1e-6
    import ctypes
    fp = ctypes.POINTER(ctypes.c_float)
    assert abs(oracle_clib.oracle_sym_rel_err(a.ctypes.data_as(fp), b.ctypes.data_as(fp), 4) - e) < 1e-6


def test_flop_model():
    assert orc.attn_flops(8, 32, 4096, 128, True) == pytest.approx(1.0995e12, rel=1e-4)
    assert orc.attn_flops(4, 8, 1024, 64, False) == pytest.approx(8.590e9, rel=1e-4)
    assert math.isclose(orc.attn_bytes(8, 32, 4096, 128, lse=False), 1073.7e6, rel_tol=1e-4)
