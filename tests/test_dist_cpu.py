"""CPU, world_size=2 over gloo: the N>1 path (contiguous (batch x head) sharding + one
all-gather) reproduces the single-process result exactly.  The HIP kernel cannot run here,
so the oracle's SDPA stands in as the per-shard compute (test-only injection)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import attn_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gqa_oracle(q, k, v, causal):
    """The oracle on a grouped layout: every query head reads key/value head h // G (expanded, then sdpa_reference)."""
    G = q.shape[1] // k.shape[1]
    return orc.sdpa_oracle(q, k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1), causal)


def _worker(rank, world, port, shape, hkv, causal, bshd, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_impls_amd.dist import flash_attn_sharded, shard_bounds
        torch.manual_seed(0)
        B, H, S, D = shape
        if bshd:                # (B,S,H,D) storage viewed as (B,H,S,D): the (batch, head) axes do not flatten in place
            q = torch.randn(B, S, H, D).transpose(1, 2)
            k, v = (torch.randn(B, S, hkv, D).transpose(1, 2) for _ in range(2))
        else:
            q = torch.randn(B, H, S, D)
            k, v = (torch.randn(B, hkv, S, D) for _ in range(2))
        full = flash_attn_sharded(q, k, v, causal, attn_fn=_gqa_oracle)
        ref = _gqa_oracle(q, k, v, causal)
        lo, hi = shard_bounds(B * hkv, rank, world)
        ok = torch.equal(full, ref) and full.shape == (B, H, S, D) and hi > lo
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,hkv,causal,bshd", [((2, 2, 32, 16), 2, True, False),
                                                   ((1, 3, 20, 16), 3, False, False),
                                                   ((3, 4, 24, 16), 1, True, False),       # multi-query: 3 units over 2 ranks
                                                   ((2, 6, 16, 16), 3, False, True)])      # grouped, strided storage
def test_sharded_equals_single_process_world2(shape, hkv, causal, bshd):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, shape, hkv, causal, bshd, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_local_shard_views_and_copies():
    from flash_attention_impls_amd.dist import local_shard, shard_bounds
    B, H, S, D = 3, 4, 5, 8
    t = torch.arange(B * H * S * D, dtype=torch.float32).reshape(B, H, S, D)
    flat = t.reshape(B * H, 1, S, D)
    for world in (1, 2, 5, 16):
        got = [local_shard(t, r, world) for r in range(world)]
        assert torch.equal(torch.cat(got), flat)
        assert all(g.data_ptr() == t.data_ptr() + 4 * S * D * shard_bounds(B * H, r, world)[0] for r, g in enumerate(got) if g.numel())
    # grouped units and a layout whose (batch, head) axes cannot be flattened without a copy
    ts = torch.arange(B * H * S * D, dtype=torch.float32).reshape(B, S, H, D).transpose(1, 2)
    for world in (1, 2, 3, 7):
        got = torch.cat([local_shard(ts, r, world, group=2) for r in range(world)])
        assert torch.equal(got, ts.reshape(B * H // 2, 2, S, D))
    with pytest.raises(ValueError):
        local_shard(t, 0, 1, group=3)
