"""Diagnostic: board power and clocks (rocm-smi) while one workload runs back to back in this process.

  python tools/power_probe.py [--causal 1] [--zeros 0] [--seconds 6] [--lib build/libX.so]
Prints the achieved TFLOP/s and the rocm-smi samples taken during the run.  Developer tool: answers whether a kernel is
running against the board's power limit (then removing stalls buys nothing and removing work does)."""
import argparse
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import importlib

fa_mod = importlib.import_module("flash_attention_impls_amd.flash_attn")
from flash_attention_impls_amd.bench_utils import attn_flops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--zeros", type=int, default=0)
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--lib", default=None)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
if a.lib:
    fa_mod._lib_handle = fa_mod.load_library(a.lib)
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp8": torch.float8_e4m3fn}[a.dtype]
mk = (lambda: torch.zeros(a.B, a.H, a.S, a.D, device="cuda").to(DT)) if a.zeros else (lambda: torch.randn(a.B, a.H, a.S, a.D, device="cuda").to(DT))
q, k, v = mk(), mk(), mk()
samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out.strip()))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), f"rocm-smi failed: {e}"))
        time.sleep(0.5)


for _ in range(5):
    fa_mod.flash_attn(q, k, v, bool(a.causal))
torch.cuda.synchronize()
t = threading.Thread(target=poll)
t.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < a.seconds:
    for _ in range(50):
        fa_mod.flash_attn(q, k, v, bool(a.causal))
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
t.join()
ms = e0.elapsed_time(e1) / n
print(f"{'zeros' if a.zeros else 'randn'} causal={a.causal}: {ms:.4f} ms per launch = {attn_flops(a.B, a.H, a.S, a.D, bool(a.causal)) / ms / 1e9:.1f} TFLOP/s over {n} launches")
for ts, s in samples[:: max(1, len(samples) // 6)]:
    print(f"--- t+{ts - t0:.1f}s\n{s}")
