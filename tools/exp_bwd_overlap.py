"""Experiment: overlap the HBM-bound dQ GEMM of batch chunk c with the matrix-bound dK/dV kernel of chunk c+1 on a second stream
(the two cannot share a CU, so any gain comes from the GEMM needing fewer CUs than the whole chip to saturate HBM).
Needs an experiment build of the library: bash tools/build_variant.sh exp -DFA_BWD_EXPERIMENTS, whose switch
FA_MI355_BWD_PHASE selects 1 = pre-pass + dK/dV kernel, 2 = dQ GEMM alone (the shipped library ignores the variable)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

lib = fmod.load_library(os.path.join("build", "libexp.so"))
B, H, S, D, causal = 8, 32, 4096, 128, True
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(4))
scale = D ** -0.5
o, lse = fmod._fwd_raw(lib, q, k, v, causal, scale, None, True)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
code = fmod._dtype_code(q.dtype)


def call(b0, b1, ws, stream, phase):
    os.environ["FA_MI355_BWD_PHASE"] = str(phase)
    t = [x[b0:b1] for x in (q, k, v, o, do, lse, dq, dk, dv)]
    rc = lib.fa_bwd_ex(*[x.data_ptr() for x in t], b1 - b0, H, H, S, S, D, *[None] * 8, code, 1 if causal else 0, scale,
                       ws.data_ptr(), ws.numel(), stream.cuda_stream)
    assert rc == 0, lib.fa_last_error().decode()


def timed(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[1]


main = torch.cuda.current_stream()
ws_full = torch.empty(lib.fa_bwd_ds_workspace_bytes(B, H, H, S, S, D), dtype=torch.uint8, device="cuda")
t_serial = timed(lambda: call(0, B, ws_full, main, 0))
ref = [x.clone() for x in (dq, dk, dv)]
fl = 2.5 * attn_flops(B, H, S, D, causal)
print(f"serial, one launch set over the batch: {t_serial:.4f} ms  {fl / t_serial / 1e9:.1f} TF/s")
for n in (2, 4, 8):
    bc = B // n
    wss = [torch.empty(lib.fa_bwd_ds_workspace_bytes(bc, H, H, S, S, D), dtype=torch.uint8, device="cuda") for _ in range(n)]
    t_chunks = timed(lambda: [call(c * bc, (c + 1) * bc, wss[c], main, 0) for c in range(n)])
    for prio_b in (0, -1):
        s2 = torch.cuda.Stream(priority=prio_b)
        evs = [torch.cuda.Event() for _ in range(n)]
        done = torch.cuda.Event()

        def overlapped():
            s2.wait_stream(main)
            for c in range(n):
                call(c * bc, (c + 1) * bc, wss[c], main, 1)
                evs[c].record(main)
                s2.wait_event(evs[c])
                call(c * bc, (c + 1) * bc, wss[c], s2, 2)
            done.record(s2)
            main.wait_event(done)
        t = timed(overlapped)
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref, (dq, dk, dv)))
        print(f"{n} chunks: serial chunks {t_chunks:.4f} ms | GEMM on a second stream (priority {prio_b}) {t:.4f} ms  {fl / t / 1e9:.1f} TF/s  "
              f"({100 * (t_serial / t - 1):+.1f} % vs serial)  bitwise equal: {same}", flush=True)
os.environ.pop("FA_MI355_BWD_PHASE", None)
