"""Diagnostic: where the waves of the dK/dV kernel spend their cycles (build with tools/build_variant.sh bwdstamp -DFA_BWD_STAMP;
run on the GPU box: FA_MI355_LIB=build/libbwdstamp.so python tools/stamps_bwd.py [--causal 0]).  Sums over all workgroups of ONE
backward launch, per wave: waves 0-3 are the score waves, 4-7 the gradient waves."""
import argparse, ctypes, importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
fmod = importlib.import_module("flash_attention_impls_amd.flash_attn")
ap = argparse.ArgumentParser()
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--B", type=int, default=8); ap.add_argument("--H", type=int, default=32); ap.add_argument("--S", type=int, default=4096)
a = ap.parse_args()
lib = fmod.load_library()
lib.fa_debug_read_bwd_stamps.restype = ctypes.c_int
lib.fa_debug_read_bwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
q, k, v, do = (torch.randn(a.B, a.H, a.S, 128, device="cuda").to(torch.bfloat16) for _ in range(4))
scale = 128 ** -0.5
o, lse = fmod._fwd_raw(lib, q, k, v, bool(a.causal), scale, None, True)
for _ in range(3):
    fmod._bwd_raw(lib, q, k, v, o, lse, do, bool(a.causal), scale)
torch.cuda.synchronize()
buf = np.zeros((8, 8), dtype=np.uint64)
lib.fa_debug_read_bwd_stamps(buf.ctypes.data, buf.nbytes, 1)           # clear
fmod._bwd_raw(lib, q, k, v, o, lse, do, bool(a.causal), scale)
torch.cuda.synchronize()
assert lib.fa_debug_read_bwd_stamps(buf.ctypes.data, buf.nbytes, 0) == 0
print(f"# fa_bwd_dkdv_kernel stamps, (B,H,S,D) = {(a.B, a.H, a.S, 128)} bf16 causal={a.causal}: sums over all workgroups of one launch (s_memtime cycles)")
print("# the stamps perturb the kernel (they drain the LDS reads it keeps in flight): read the shares, not the lengths")
print(f"{'wave':>4s} {'role':9s} {'whole kernel':>14s} {'steps':>9s} {'cycles/step':>12s} | shares of the kernel: {'work':>6s} {'DMA wait':>9s} {'barrier':>8s} {'staging':>8s} {'prologue':>9s} {'rest':>6s}")
for w in range(8):
    tot, dw, bar, iss, work, steps, pro = (float(buf[w][i]) for i in (0, 1, 2, 3, 4, 5, 6))
    if tot == 0:
        continue
    rest = tot - (dw + bar + iss + work + pro)
    print(f"{w:4d} {'score' if w < 4 else 'gradient':9s} {tot:14.4g} {steps:9.0f} {(dw + bar + iss + work) / max(steps, 1):12.0f} | "
          f"{'':21s} {work / tot:6.3f} {dw / tot:9.3f} {bar / tot:8.3f} {iss / tot:8.3f} {pro / tot:9.3f} {rest / tot:6.3f}")
