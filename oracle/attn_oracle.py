"""CPU oracle for the FlashAttention-forward hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path
(``flash_attention_impls_amd``) never imports anything under ``oracle/`` and
fails loudly when its HIP library is missing.

What is restated here (reference = santiweide/flash-attention-impls @ 2025-10-31):

* ``sdpa_oracle``            <- ``sdpa_reference``  code/triton_fa2/FA2-triton.py:311-323
  (fp32 ``scaled_dot_product_attention`` on the dtype-rounded inputs,
  ``scale = 1/sqrt(D)`` :314, result cast back to the input dtype :323).
* ``tiled_online_softmax``   <- ``_fwd_kernel``     code/triton_fa2/FA2-triton.py:56-93
  (split-Q loop over KV tiles, causal rule ``col > row -> -inf`` :70-73,
  running max ``m`` and normaliser ``l`` :75-77, normalise-every-step
  accumulate :79-82, stores O, m, l :87-93).  The build's kernel uses deferred
  normalisation; ``deferred=True`` restates that variant (mathematically equal).
* ``naive_attention_f64``    <- ``attention_baseline_kernel``
  code/cutlass_cuda_fa1/run/test_flash_attn.cu:548-615 (3-pass, one query row at
  a time: scores, max, exp/sum, weighted V sum), in float64.
* ``sym_rel_err``            <- ``compute_max_relative_error``
  code/cutlass_cuda_fa1/run/test_flash_attn.cu:108-143 (|a-b|/(|a|+|b|+1e-5)).
* ``naive_attention_bwd_f64`` / ``tiled_recompute_bwd`` <- ``_bwd_kernel``
  code/triton_fa2/FA2-triton.py:98-170 (recompute p from the saved m, l :156;
  dv = p^T dO :158, dP = dO v^T :159, dp_sum = rowsum(dP*p) :160,
  dS = (dP - dp_sum*p)*scale :161, dQ += dS k :163, dK = dS^T q :164), and
  ``sdpa_bwd_oracle`` <- autograd through ``sdpa_reference`` (what the reference's
  ``fwd_bwd`` harness differentiates, :357-372).

Parity pin: ``tests/golden/*.npz`` were produced by ``oracle/gen_golden.py``, which
imports the reference's own ``sdpa_reference`` (and, for D<=64, runs its Triton
``_fwd_kernel`` under TRITON_INTERPRET=1 for m/l) in the build container.
``tests/test_oracle.py`` checks every function here against those vectors.
"""
from __future__ import annotations

import math

import numpy as np

try:  # torch is only needed by sdpa_oracle / cpu timing
    import torch
    import torch.nn.functional as F
except Exception:  # pragma: no cover
    torch = None


# --------------------------------------------------------------------------
# dtype helpers (bf16 / fp16 / fp8-e4m3fn rounding in numpy, bit exact vs torch)
# --------------------------------------------------------------------------
def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 bit pattern (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def round_to_dtype(x: np.ndarray, dtype: str) -> np.ndarray:
    """Return fp32 array holding x rounded to ``dtype`` ('fp32','fp16','bf16')."""
    if dtype == "fp32":
        return np.asarray(x, dtype=np.float32)
    if dtype == "fp16":
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        return bf16_bits_to_f32(f32_to_bf16_bits(x))
    raise ValueError(dtype)


# --------------------------------------------------------------------------
# sdpa_reference restatement   (FA2-triton.py:311-323)
# --------------------------------------------------------------------------
def sdpa_oracle(q, k, v, causal: bool = False, scale: float | None = None):
    """fp32 SDPA on CPU copies of (B,H,N,D) torch tensors; returns q.dtype."""
    B, H, N, D = q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)                       # :314
    q2 = q.reshape(B * H, N, D).to(torch.float32)        # :315-317
    k2 = k.reshape(B * H, N, D).to(torch.float32)
    v2 = v.reshape(B * H, N, D).to(torch.float32)
    out = F.scaled_dot_product_attention(                # :320-322
        q2, k2, v2, attn_mask=None, is_causal=causal, scale=scale)
    return out.reshape(B, H, N, D).to(q.dtype)           # :323


# --------------------------------------------------------------------------
# attention_baseline_kernel restatement in float64 (test_flash_attn.cu:548-615)
# --------------------------------------------------------------------------
def naive_attention_f64(q, k, v, causal: bool = False, scale: float | None = None):
    """q,k,v: numpy (B,H,N,D) any float dtype. Returns (o f64, lse f64).
    k, v may have another length N_k (not a reference case: its q, k, v share one N); the causal mask is then
    bottom-right aligned, key j visible to query i iff j <= i + N_k - N (for N_k == N the reference's rule); for
    N_k < N the first N - N_k queries see no key: their output is 0 and their LSE -inf (the l == 0 guard of
    flash_attn_cutlass.cu:446-452)."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    B, H, N, D = q.shape
    Nk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    s = np.einsum("bhid,bhjd->bhij", q, k) * scale       # pass 1 :575-586
    if causal:
        i = np.arange(N)[:, None]
        j = np.arange(Nk)[None, :]
        s = np.where(j > i + (Nk - N), -np.inf, s)       # FA2-triton.py:70-73
    m = s.max(axis=-1, keepdims=True)                    # :588-592
    m = np.where(np.isneginf(m), 0.0, m)                 # a query that sees no key (only possible for N_k < N, causal)
    p = np.exp(s - m)                                    # pass 2 :594-600
    l = p.sum(axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        o = np.where(l > 0, np.einsum("bhij,bhjd->bhid", p, v) / l, 0.0)   # pass 3 :602-613; l == 0 -> 0 (flash_attn_cutlass.cu:446-452)
        lse = np.where(l > 0, m + np.log(l), -np.inf)[..., 0]
    return o, lse


# --------------------------------------------------------------------------
# _fwd_kernel restatement (FA2-triton.py:56-93), fp32, tile by tile
# --------------------------------------------------------------------------
def tiled_online_softmax(q, k, v, causal: bool = False, scale: float | None = None,
                         block_m: int = 128, block_n: int = 128,
                         deferred: bool = False, skip_masked_tiles: bool = False,
                         p_dtype: str | None = None):
    """fp32 tile loop.  q,k,v numpy (B,H,N,D) fp32 (already dtype-rounded).

    deferred=False : reference arithmetic (acc normalised every step :79-82).
    deferred=True  : the build's arithmetic (acc rescaled by exp(m_old-m_new),
                     single divide by l at the end) + optional causal tile skip.
    p_dtype        : round P to 'bf16'/'fp16' before the PV product, as the MFMA
                     path does (None = keep fp32 like the reference's tl.dot :79).
    Returns (o fp32, m fp32 (B,H,N), l fp32 (B,H,N)).
    """
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    B, H, N, D = q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)                                   # :183
    scale = np.float32(scale)
    o = np.zeros((B, H, N, D), np.float32)
    m_out = np.zeros((B, H, N), np.float32)
    l_out = np.zeros((B, H, N), np.float32)
    for b in range(B):
        for h in range(H):
            for row_start in range(0, N, block_m):                   # program_id(1) :41
                rows = np.arange(row_start, min(row_start + block_m, N))
                qt = q[b, h, rows]                                   # :52-53
                m_i = np.full(len(rows), -np.inf, np.float32)        # :56
                l_i = np.zeros(len(rows), np.float32)                # :57
                acc = np.zeros((len(rows), D), np.float32)           # :58
                for col_start in range(0, N, block_n):               # :60
                    if causal and skip_masked_tiles and col_start > rows[-1]:
                        break
                    cols = np.arange(col_start, min(col_start + block_n, N))
                    kt = k[b, h, cols]                               # :63-66
                    vt = v[b, h, cols]
                    qk = (qt @ kt.T).astype(np.float32) * scale      # :68
                    if causal:
                        qk = np.where(cols[None, :] > rows[:, None],
                                      np.float32(-np.inf), qk)       # :70-73
                    m_ij = np.maximum(m_i, qk.max(axis=1))           # :75
                    p = np.exp(qk - m_ij[:, None]).astype(np.float32)  # :76
                    corr = np.exp(m_i - m_ij).astype(np.float32)
                    l_ij = l_i * corr + p.sum(axis=1)                # :77
                    if p_dtype is not None:
                        p = round_to_dtype(p, p_dtype)
                    pv = (p @ vt).astype(np.float32)                 # :79
                    if deferred:
                        acc = acc * corr[:, None] + pv
                    else:
                        alpha = (l_i * corr) / l_ij                  # :80
                        beta = np.float32(1.0) / l_ij                # :81
                        acc = acc * alpha[:, None] + pv * beta[:, None]  # :82
                    m_i = m_ij                                       # :84
                    l_i = l_ij                                       # :85
                if deferred:
                    # final normalise with l==0 guard (flash_attn_cutlass.cu:446-452)
                    inv = np.where(l_i > 0, np.float32(1.0) / np.where(l_i > 0, l_i, 1), 0)
                    acc = acc * inv[:, None]
                o[b, h, rows] = acc                                  # :87-88
                m_out[b, h, rows] = m_i                              # :90-93
                l_out[b, h, rows] = l_i
    return o, m_out, l_out


# --------------------------------------------------------------------------
# _bwd_kernel restatement (FA2-triton.py:98-170)
# --------------------------------------------------------------------------
def naive_attention_bwd_f64(q, k, v, do, causal: bool = False, scale: float | None = None):
    """Closed-form float64 gradients of o = softmax(scale q k^T [+mask]) v.
    q,k,v,do: numpy (B,H,N,D).  Returns (dq, dk, dv, delta) in float64, delta = rowsum(do*o).
    k, v may have another length N_k (bottom-right aligned causal mask, as in naive_attention_f64)."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    do = np.asarray(do, dtype=np.float64)
    B, H, N, D = q.shape
    Nk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    s = np.einsum("bhid,bhjd->bhij", q, k) * scale                   # :147
    if causal:
        i = np.arange(N)[:, None]
        j = np.arange(Nk)[None, :]
        s = np.where(j > i + (Nk - N), -np.inf, s)                   # :148-151
    m = s.max(axis=-1, keepdims=True)
    m = np.where(np.isneginf(m), 0.0, m)                             # a query that sees no key: P = 0, no gradient
    p = np.exp(s - m)
    l = p.sum(axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(l > 0, p / l, 0.0)                              # :156 (exp(qk-m)/l)
    dv = np.einsum("bhij,bhid->bhjd", p, do)                         # :158
    dp = np.einsum("bhid,bhjd->bhij", do, v)                         # :159
    delta = (dp * p).sum(axis=-1, keepdims=True)                     # :160
    ds = (dp - delta) * p * scale                                    # :161
    dq = np.einsum("bhij,bhjd->bhid", ds, k)                         # :163
    dk = np.einsum("bhij,bhid->bhjd", ds, q)                         # :164
    return dq, dk, dv, delta[..., 0]


def tiled_recompute_bwd(q, k, v, do, lse, causal: bool = False, scale: float | None = None,
                        block_m: int = 128, block_n: int = 128, w_dtype: str | None = None,
                        delta=None):
    """fp32 tile loop of the reference backward: one program per BLOCK_M query rows sweeping all
    BLOCK_N key tiles (:119-170), p recomputed from the saved statistics (here lse = m + ln l, so
    p = exp(qk - lse), equal to exp(qk - m)/l :156).  REFERENCE DEFECT (not reproduced): :160-161 form
    dS = (dP - rowsum_tile(dP*p) * p) * scale; the softmax Jacobian is dS = p * (dP - rowsum(dP*p)) * scale
    with the sum over the whole row.  The interpreted reference kernel's dV equals autograd through
    sdpa_reference, its dQ/dK do not (oracle/gen_golden.py prints both; tests/golden/bwd_fp16_d64_*.npz keep
    them as *_kernel).  The restatement uses the correct form with delta = rowsum(do*o) (FlashAttention-2),
    which is what autograd of the reference's own sdpa_reference yields.
    w_dtype: round p and dS to 'bf16'/'fp16' before the three gradient products, as the MFMA path does.
    Returns (dq, dk, dv) fp32."""
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    do = np.asarray(do, dtype=np.float32)
    lse = np.asarray(lse, dtype=np.float32)
    B, H, N, D = q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    scale = np.float32(scale)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    for b in range(B):
        for h in range(H):
            if delta is None:
                # full-row softmax output in fp32 for delta
                s_full = (q[b, h] @ k[b, h].T) * scale
                if causal:
                    s_full = np.where(np.arange(N)[None, :] > np.arange(N)[:, None], np.float32(-np.inf), s_full)
                o_full = np.exp(s_full - lse[b, h][:, None]).astype(np.float32) @ v[b, h]
                dl = (do[b, h] * o_full).sum(axis=1).astype(np.float32)
            else:
                dl = np.asarray(delta, dtype=np.float32)[b, h]
            for row_start in range(0, N, block_m):                           # program_id(1) :121
                rows = np.arange(row_start, min(row_start + block_m, N))
                qt, dot = q[b, h, rows], do[b, h, rows]                      # :131-134
                acc = np.zeros((len(rows), D), np.float32)                   # :141
                for col_start in range(0, N, block_n):                       # :144
                    cols = np.arange(col_start, min(col_start + block_n, N))
                    kt, vt = k[b, h, cols], v[b, h, cols]
                    qk = (qt @ kt.T).astype(np.float32) * scale              # :147
                    if causal:
                        qk = np.where(cols[None, :] > rows[:, None], np.float32(-np.inf), qk)   # :148-151
                    p = np.exp(qk - lse[b, h, rows][:, None]).astype(np.float32)                 # :156
                    dp = (dot @ vt.T).astype(np.float32)                     # :159
                    ds = (dp - dl[rows][:, None]) * p * scale                # :160-161
                    if w_dtype is not None:
                        p = round_to_dtype(p, w_dtype)
                        ds = round_to_dtype(ds, w_dtype)
                    dv[b, h, cols] += (p.T @ dot).astype(np.float32)         # :158,166-167
                    acc += (ds @ kt).astype(np.float32)                      # :163
                    dk[b, h, cols] += (ds.T @ qt).astype(np.float32)         # :164,168-169
                dq[b, h, rows] = acc                                         # :171-172
    return dq, dk, dv


def sdpa_bwd_oracle(q, k, v, do, causal: bool = False, scale: float | None = None):
    """Autograd through the sdpa_reference restatement (fp32 math on CPU copies of the dtype-rounded
    tensors).  Returns (o, dq, dk, dv) as fp32 torch tensors."""
    B, H, N, D = q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    q2, k2, v2 = [t.detach().to("cpu", torch.float32).clone().requires_grad_(True) for t in (q, k, v)]
    o = F.scaled_dot_product_attention(q2, k2, v2, is_causal=causal, scale=scale)
    o.backward(do.detach().to("cpu", torch.float32))
    return o.detach(), q2.grad, k2.grad, v2.grad


# --------------------------------------------------------------------------
# compute_max_relative_error restatement (test_flash_attn.cu:108-143)
# --------------------------------------------------------------------------
def sym_rel_err(a, b) -> float:
    a = np.asarray(a, dtype=np.float32).ravel()
    b = np.asarray(b, dtype=np.float32).ravel()
    return float(np.max(np.abs(a - b) / (np.abs(a) + np.abs(b) + np.float32(1e-5))))


# --------------------------------------------------------------------------
# FLOP / byte model (test_flash_attn.cu:308-320; report/pmph-a6.tex:155-174)
# --------------------------------------------------------------------------
def attn_flops(B, H, S, D, causal: bool) -> float:
    f = 4.0 * B * H * S * S * D                      # test_flash_attn.cu:308
    return f / 2 if causal else f                    # FA2 convention


def attn_bwd_flops(B, H, S, D, causal: bool) -> float:
    """Five N x N x D products (S, dP, dV, dK, dQ) = 2.5 x the forward (FlashAttention-2 convention)."""
    return 2.5 * attn_flops(B, H, S, D, causal)


def attn_bytes(B, H, S, D, in_bytes=2, out_bytes=2, lse=True) -> float:
    n = B * H * S * D
    return 3 * n * in_bytes + n * out_bytes + (B * H * S * 4 if lse else 0)


# --------------------------------------------------------------------------
# CPU baseline timing helper (bench.py cpu_baseline leg): the reference's CPU-
# runnable path is sdpa_reference == torch CPU SDPA.
# --------------------------------------------------------------------------
def time_cpu_sdpa(B, H, S, D, dtype, causal, iters=1, seed=0, threads=None):
    import os
    import time
    if threads is None:
        threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g).to(dtype)
    k = torch.randn(B, H, S, D, generator=g).to(dtype)
    v = torch.randn(B, H, S, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    with torch.no_grad():
        F.scaled_dot_product_attention(q[:1, :1], k[:1, :1], v[:1, :1], is_causal=causal, scale=scale)
        t0 = time.perf_counter()
        for _ in range(iters):
            F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
        dt = (time.perf_counter() - t0) / iters
    return dt, threads
