"""Multi-GPU sharding of the attention forward: one process per GPU, (batch x head) units
split contiguously over ranks, no data-path collective, one final all-gather of O over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in CPU tests).

The reference has no distributed code at all; each (b, h) is an independent
problem exactly as in its grid (code/triton_fa2/FA2-triton.py:40-43), which is what makes
the split exact: the gathered O is bitwise the single-GPU O.

With grouped key/value heads (``k``, ``v`` with H_kv < H heads) the independent unit is a (batch, key/value head)
pair together with the G = H / H_kv query heads that read it: the split runs over those B * H_kv units, so that a
rank's queries always find their keys and values in the rank's own shard.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(units: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) slice of ``units`` owned by ``rank``; the first ``units % world`` ranks get one extra unit."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def local_shard(t: torch.Tensor, rank: int, world: int, group: int = 1) -> torch.Tensor:
    """The units of a (B, H, S, D) tensor owned by ``rank`` as (n, group, S, D): a unit is ``group`` consecutive
    heads of one batch (group = 1 for keys / values, G = H / H_kv for the queries of a grouped-query layout).
    A view whenever the (batch, head) axes can be flattened in place; otherwise only the rank's own units are copied
    (never the whole tensor: slice first, then make contiguous)."""
    B, H, S, D = t.shape
    if group <= 0 or H % group != 0:
        raise ValueError(f"{H} heads are not a multiple of the unit size {group}")
    hu = H // group
    lo, hi = shard_bounds(B * hu, rank, world)
    tu = t.unflatten(1, (hu, group))                     # (B, hu, group, S, D), always a view
    try:
        return tu.view(B * hu, group, S, D)[lo:hi]
    except RuntimeError:                                 # (batch, head) strides do not flatten (e.g. a (B,S,H,D) buffer)
        parts = []
        for b in range(lo // hu, (hi + hu - 1) // hu if hi > lo else lo // hu):
            u0, u1 = max(lo, b * hu) - b * hu, min(hi, (b + 1) * hu) - b * hu
            parts.append(tu[b, u0:u1])
        if not parts:
            return t.new_empty((0, group, S, D))
        return torch.cat(parts, dim=0)


def gather_output(o_local: torch.Tensor, units: int, group=None) -> torch.Tensor:
    """All-gather the per-rank (n_r, G, S, D) outputs into (units, G, S, D) on every rank.
    Equal shards use one all_gather_into_tensor (one direct message per peer link on xGMI);
    ragged shards are padded to the largest shard and trimmed."""
    world = dist.get_world_size(group)
    n_max = -(-units // world)
    tail = tuple(o_local.shape[1:])
    send = o_local.contiguous()
    if send.shape[0] != n_max:
        pad = torch.zeros((n_max - send.shape[0],) + tail, dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    full = torch.empty((world * n_max,) + tail, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(full, send, group=group)
    if units % world == 0:
        return full
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(units, r, world)
        parts.append(full[r * n_max: r * n_max + (hi - lo)])
    return torch.cat(parts, dim=0)


def _out_dtype(q: torch.Tensor) -> torch.dtype:
    """Element type of O for inputs of q's type: fp8 inputs give a bf16 output (fa_fwd_fp8), 16-bit inputs their own, fp32
    inputs fp32 (the operator computes those in fp16 and converts back, FA2-triton.py:241-244)."""
    return torch.bfloat16 if q.dtype == torch.float8_e4m3fn else q.dtype


def flash_attn_sharded(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False, *,
                       attn_fn: Callable | None = None, group=None, gather: bool = True,
                       rank: int | None = None, world: int | None = None, always_collective: bool = False,
                       **attn_kwargs):
    """Every rank holds the full inputs (or at least its own units); each computes its contiguous slice of the
    B * H_kv units and, if ``gather``, all ranks receive the full (B, H, S, D) output.

    ``attn_fn`` defaults to the HIP ``flash_attn``; CPU tests inject an oracle here to exercise the sharding/gather
    logic over gloo.  ``rank`` / ``world`` default to the process group's; passing them (with ``gather=False``) computes
    any rank's shard in a single process, which is how the GPU tests check the split against the unsharded launch.
    ``attn_kwargs`` (``descale=...`` for fp8 inputs, ``softmax_scale=...``) are passed through to ``attn_fn``.
    ``always_collective``: run the all-gather even at world size 1 (a one-rank RCCL communicator is valid: this is how a
    one-GPU box executes the collective path for real).
    """
    if attn_fn is None:
        from .flash_attn import flash_attn as attn_fn
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, H, S, D = q.shape
    Hkv = k.shape[1]
    if v.shape[1] != Hkv or Hkv <= 0 or H % Hkv != 0:
        raise ValueError(f"q has {H} heads, k / v have {k.shape[1]} / {v.shape[1]}: need H % H_kv == 0")
    G = H // Hkv
    ql = local_shard(q, rank, world, G)
    kl, vl = local_shard(k, rank, world), local_shard(v, rank, world)
    # a rank without units still takes part in the gather: its (empty) shard must have the OUTPUT's element type -- for fp8
    # inputs that is bf16, not q's -- or the ranks' buffers of all_gather_into_tensor disagree in type and size
    o_local = attn_fn(ql, kl, vl, causal, **attn_kwargs) if ql.shape[0] > 0 else ql.new_empty(ql.shape, dtype=_out_dtype(q))
    if not gather:
        return o_local
    if world == 1 and not (always_collective and dist.is_initialized()):
        return o_local.reshape(B, H, S, D)
    return gather_output(o_local, B * Hkv, group).reshape(B, H, S, D)
