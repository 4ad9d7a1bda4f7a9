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


def _worker(rank, world, port, shape, causal, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_impls_amd.dist import flash_attn_sharded, shard_bounds
        torch.manual_seed(0)
        B, H, S, D = shape
        q, k, v = (torch.randn(B, H, S, D) for _ in range(3))
        full = flash_attn_sharded(q, k, v, causal, attn_fn=lambda a, b, c, cz: orc.sdpa_oracle(a, b, c, cz))
        ref = orc.sdpa_oracle(q, k, v, causal)
        lo, hi = shard_bounds(B * H, rank, world)
        ok = torch.equal(full, ref) and full.shape == (B, H, S, D) and hi > lo
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,causal", [((2, 2, 32, 16), True), ((1, 3, 20, 16), False)])
def test_sharded_equals_single_process_world2(shape, causal):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, shape, causal, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
