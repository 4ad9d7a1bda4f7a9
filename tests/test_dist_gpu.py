"""The sharding code on the HIP path: `dist.flash_attn_sharded` with the real kernel at world size 1,
every rank of a W-way split emulated in one process (local_shard -> flash_attn -> concatenation) bitwise against the
unsharded launch, a two-rank gloo rehearsal on one GPU, and `bench.py --gpus 2` started from a plain shell."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402
from flash_attention_impls_amd.dist import flash_attn_sharded, local_shard, shard_bounds  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _qkv(B, H, Hkv, S, D, dt=torch.bfloat16, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g).to(dt).cuda()
    k, v = (torch.randn(B, Hkv, S, D, generator=g).to(dt).cuda() for _ in range(2))
    return q, k, v


@pytest.mark.parametrize("B,H,Hkv,S,D,causal", [(2, 4, 4, 512, 128, True), (1, 5, 5, 300, 128, False),
                                                (2, 8, 2, 384, 64, True), (3, 2, 1, 260, 128, True)])
def test_sharded_world1_and_emulated_ranks(B, H, Hkv, S, D, causal):
    q, k, v = _qkv(B, H, Hkv, S, D)
    ref = fa.flash_attn(q, k, v, causal)
    assert torch.equal(flash_attn_sharded(q, k, v, causal), ref)              # world size 1, no process group
    for W in (2, 3, 8):                                                        # (B * H_kv = 5 or 3: ragged and empty shards)
        parts = [flash_attn_sharded(q, k, v, causal, gather=False, rank=r, world=W) for r in range(W)]
        sizes = [p.shape[0] for p in parts]
        assert sizes == [shard_bounds(B * Hkv, r, W)[1] - shard_bounds(B * Hkv, r, W)[0] for r in range(W)]
        got = torch.cat(parts, dim=0).reshape(B, H, S, D)
        assert torch.equal(got, ref), (W, sizes)


def test_sharded_strided_inputs_copy_only_the_shard():
    """(B,S,H,D) storage viewed as (B,H,S,D): the shard is sliced first, the kernel reads it through its strides."""
    B, H, S, D = 2, 6, 320, 128
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).cuda().transpose(1, 2) for _ in range(3))
    ref = fa.flash_attn(q, k, v, True)
    parts = [flash_attn_sharded(q, k, v, True, gather=False, rank=r, world=4) for r in range(4)]
    assert torch.equal(torch.cat(parts, dim=0).reshape(B, H, S, D), ref)
    assert local_shard(q, 1, 4).shape == (3, 1, S, D)


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_fp8_shards_with_fewer_units_than_ranks():
    """fp8 inputs give a bf16 output: a rank without units must return an empty bf16 shard (q's type would make the ranks'
    all-gather buffers disagree), and the per-tensor scales reach the kernel through the sharded entry."""
    B, H, S, D = 1, 3, 256, 128
    g = torch.Generator().manual_seed(5)
    f32 = [torch.randn(B, H, S, D, generator=g) for _ in range(3)]
    dsc = tuple(float(t.abs().max()) / 448.0 for t in f32)
    q, k, v = [(t / s_).to(torch.float8_e4m3fn).cuda() for t, s_ in zip(f32, dsc)]
    ref = fa.flash_attn(q, k, v, False, descale=dsc)
    parts = [flash_attn_sharded(q, k, v, False, gather=False, rank=r, world=8, descale=dsc) for r in range(8)]
    assert [p.shape[0] for p in parts] == [1, 1, 1, 0, 0, 0, 0, 0]
    assert all(p.dtype == torch.bfloat16 for p in parts)
    assert torch.equal(torch.cat(parts, dim=0).reshape(B, H, S, D), ref)
    assert not torch.equal(ref, fa.flash_attn(q, k, v, False))          # (the scales matter)


def test_two_rank_gloo_rehearsal_on_one_gpu():
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_rehearsal_worker.py")])
    assert res.returncode == 0 and "DIST_REHEARSAL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_bench_gpus2_from_a_plain_shell():
    """`python bench.py --gpus 2 --dist-backend gloo` with no launcher: bench.py starts its ranks itself (before touching the
    GPU) and rank 0's JSON line comes through, marked as a rehearsal and without a `value` (all ranks share cuda:0)."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--settle-ms", "30", "--workload", "cfg2", "--no-cpu-baseline", "--rehearse-gather",
                          "--dist-backend", "gloo"],
                         cwd=ROOT, env=env_clean, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["rehearsal"] is True and out["value"] is None and out["rehearsal_value"] > 0     # cannot be mistaken for a measurement
    assert out["step_ms_min"] <= out["step_ms_median"] <= out["step_ms_max"]
    # the final-gather timings (blocking, compute-then-gather, chunked compute || gather) ran end to end -- over gloo here,
    # so only that the code path works is checked, not its numbers
    g = out["gather"]
    assert "error" not in g, g
    assert g["ms"] > 0 and g["compute_then_gather_ms"] > 0 and g["overlapped_total_ms"] > 0 and g["overlap_chunks"] >= 2


def test_bench_more_ranks_than_gpus_is_an_error_not_a_silent_rehearsal():
    """Default backend (auto = nccl): `--gpus 2` on a box with one GPU must fail, not print a normal-looking line."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--settle-ms", "10", "--workload", "cfg2", "--no-cpu-baseline"],
                         cwd=ROOT, env=env_clean, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "fewer GPUs than ranks" in res.stderr or "GPU(s) on this box" in res.stderr, res.stderr[-1500:]


def test_rccl_path_at_world_size_one():
    """The RCCL code path for real, on one GPU (a one-rank communicator is valid): bench.py under torch.distributed.run with
    the nccl backend -- init_process_group(device_id=...), barriers, all_gather_into_tensor on device tensors, the
    comm-stream overlap with record_stream -- and dist.flash_attn_sharded's all-gather.  What an 8-GPU node runs first."""
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                "--settle-ms", "30", "--workload", "cfg2", "--no-cpu-baseline", "--no-attainable", "--no-power",
                "--dist-backend", "nccl", "--force-dist"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "rehearsal" not in out
    g = out["gather"]
    assert "error" not in g, g
    assert g["ms"] > 0 and g["compute_then_gather_ms"] > 0 and g["overlapped_total_ms"] > 0
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_rehearsal_worker.py"), "--backend", "nccl"])
    assert res.returncode == 0 and "DIST_REHEARSAL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_bench_strong_scaling_over_rccl_at_world_size_one():
    """`bench.py --scaling strong` (ONE problem split into contiguous (batch, head) units per rank -- cfg3 8/W
    batches, cfg4 16/W heads, cfg5 64/W batches) under torch.distributed.run over nccl at world size 1: the whole problem is rank
    0's shard, `value` counts the whole problem's FLOPs, and the line says "strong"."""
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                "--settle-ms", "30", "--workload", "cfg4", "--scaling", "strong", "--no-cpu-baseline", "--no-attainable", "--no-power",
                "--dist-backend", "nccl", "--force-dist"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["units_total"] == 16 and out["config"]["units_rank0"] == 16 and out["config"]["B_total"] == 1
    assert "error" not in out["gather"], out["gather"]


def test_strong_split_shards_are_the_unsharded_result():
    """The shards `bench.py --scaling strong` times -- rank r's contiguous units held as (1, n_r, S, D) -- are bitwise the
    corresponding heads of the unsharded launch, for every rank of a 2-, 4- and 8-way split of a (2, 8) problem."""
    from flash_attention_impls_amd.dist import shard_bounds
    B, H, S, D = 2, 8, 700, 128
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(B, H, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    ref = fa.flash_attn(q, k, v, True).reshape(1, B * H, S, D)
    qf, kf, vf = (t.reshape(1, B * H, S, D) for t in (q, k, v))
    for world in (2, 4, 8):
        for r in range(world):
            lo, hi = shard_bounds(B * H, r, world)
            assert torch.equal(fa.flash_attn(qf[:, lo:hi], kf[:, lo:hi], vf[:, lo:hi], True), ref[:, lo:hi])


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_few_head_shards_run_on_128_row_workgroups_and_match_bitwise(causal, dt):
    """The per-GPU shards of a strong split can be a few heads only (cfg4: 2 of 16 heads at W = 8): their 256-row grid would leave
    most CUs without a workgroup, so the host launches the 128-row form of the head_dim-128 kernel (4 waves, one per SIMD:
    fa_capi.hip rows_per_wg).  A wave does the same arithmetic in both forms: the shard must be BITWISE the slice of the full
    launch (which runs the 256-row form), here for 1-, 2- and 3-head shards of a 32-head problem."""
    import ctypes
    B, H, S, D = 1, 32, 4096, 128
    g = torch.Generator().manual_seed(41)
    q, k, v = (torch.randn(B, H, S, D, generator=g).to(dt).cuda() for _ in range(3))
    full, lse_full = fa.flash_attn(q, k, v, causal, return_lse=True)
    lib = fa.load_library()
    code = 0 if dt == torch.bfloat16 else 1
    grid, block = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.fa_fwd_launch_info(B, H, S, D, code, int(causal), ctypes.byref(grid), ctypes.byref(block), None) == 0
    assert block.value == 512                                        # the full launch: 8 waves
    for h0, n in ((0, 1), (3, 2), (29, 3)):
        assert lib.fa_fwd_launch_info(B, n, S, D, code, int(causal), ctypes.byref(grid), ctypes.byref(block), None) == 0
        assert block.value == 256, (n, grid.value, block.value)      # the shard: 4 waves
        o, lse = fa.flash_attn(q[:, h0:h0 + n], k[:, h0:h0 + n], v[:, h0:h0 + n], causal, return_lse=True)
        assert torch.equal(o, full[:, h0:h0 + n]) and torch.equal(lse, lse_full[:, h0:h0 + n])
