"""Worker of tests/test_dist_gpu.py (started by torch.distributed.run, one process per rank, all ranks on cuda:0).
Default (gloo): the HIP flash_attn computes each rank's shard, gloo assembles O on the CPU, and the result must be bitwise
the unsharded launch.  `--backend nccl` (one rank: a valid RCCL communicator on a one-GPU box): the same through
dist.flash_attn_sharded's own all_gather_into_tensor on DEVICE tensors over RCCL."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import flash_attention_impls_amd as fa  # noqa: E402
from flash_attention_impls_amd.dist import flash_attn_sharded, gather_output  # noqa: E402

nccl = "--backend" in sys.argv and sys.argv[sys.argv.index("--backend") + 1] == "nccl"
torch.cuda.set_device(0)
if nccl:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
for (B, H, Hkv, S, D, causal) in [(2, 4, 4, 300, 128, True), (1, 6, 2, 520, 64, False), (3, 1, 1, 257, 128, True)]:
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, H, S, D, generator=g).to(torch.bfloat16).cuda()
    k, v = (torch.randn(B, Hkv, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(2))
    if nccl:
        full = flash_attn_sharded(q, k, v, causal, always_collective=True).cpu()   # all-gather of device tensors over RCCL
    else:
        o_local = flash_attn_sharded(q, k, v, causal, gather=False)            # this rank's units, on the GPU
        full = gather_output(o_local.cpu(), B * Hkv).reshape(B, H, S, D)        # assembled over gloo
    ref = fa.flash_attn(q, k, v, causal).cpu()
    ok = ok and torch.equal(full, ref)
flag = torch.tensor([1 if ok else 0], device="cuda" if nccl else "cpu")
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("DIST_REHEARSAL_OK" if int(flag) == 1 else "DIST_REHEARSAL_MISMATCH", flush=True)
dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
