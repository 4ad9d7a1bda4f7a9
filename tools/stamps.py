"""Diagnostic: where a steady-state tile of the head_dim-128 forward spends its cycles, per wave (build with
tools/build_variant.sh stamp -DFA_STAMP; run on the GPU box: FA_MI355_LIB=build/libstamp.so python tools/stamps.py [--causal 0])."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import flash_attention_impls_amd as fa

ap = argparse.ArgumentParser()
ap.add_argument("--causal", type=int, default=0)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--S", type=int, default=4096)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"])
ap.add_argument("--n-wg", type=int, default=0, help="workgroups of the launch if not B*H*blocks (experiment builds with a longer grid)")
a = ap.parse_args()
lib = fa.load_library()
lib.fa_debug_read_stamps.restype = ctypes.c_int
lib.fa_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
descale = None
if a.dtype == "fp8":
    f32 = [torch.randn(a.B, a.H, a.S, 128, device="cuda") for _ in range(3)]
    descale = tuple(float(t.abs().max()) / 448.0 for t in f32)
    q, k, v = [(t / s_).to(torch.float8_e4m3fn) for t, s_ in zip(f32, descale)]
    del f32
else:
    q, k, v = (torch.randn(a.B, a.H, a.S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(30):
    fa.flash_attn(q, k, v, bool(a.causal), descale=descale)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fa.flash_attn(q, k, v, bool(a.causal), descale=descale)
e1.record()
torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) / 20 * 1e3
n_wg = min(8192, a.n_wg or a.B * a.H * ((a.S + 255) // 256) // (2 if a.causal else 1))
buf = np.zeros((n_wg, 8, 24), dtype=np.uint64)
rc = lib.fa_debug_read_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
f = buf.astype(np.float64)
tiles = f[:, :, 3]
ok = tiles > 0
n_seen = int(ok.any(axis=1).sum())
if a.n_wg:                 # experiment grid with workgroups that leave at once (no record): keep those that did work
    keep = ok.any(axis=1)
    buf, f, tiles, ok = buf[keep], f[keep], tiles[keep], ok[keep]
    n_wg = n_seen
elif n_seen < n_wg:        # persistent launch: fewer workgroups than grid slots, each walks n_wg / n_seen items (sums below are per WORKGROUP)
    print(f"persistent launch: {n_seen} workgroups for {n_wg} grid slots ({n_wg / max(1, n_seen):.1f} items per workgroup; the sums below are per workgroup, i.e. over its items)")
    buf, f, tiles, ok = buf[:n_seen], f[:n_seen], tiles[:n_seen], ok[:n_seen]
    n_wg = n_seen
print(f"workgroups {n_wg}, stamped tiles per wave: mean {tiles[ok].mean():.1f}")
if a.dtype == "fp8":
    print("wave   regions 0-3   wait+barrier+DMA issue   regions 4-7   total   [cycles per 128-key block, mean over workgroups]")
else:
    print("wave   even-block   wait+barrier   odd-block(+DMA)   total   [cycles per tile, mean over workgroups]")
for w in range(8):
    m = ok[:, w]
    e, wt, o = [(f[m, w, i] / tiles[m, w]).mean() for i in range(3)]
    print(f"{w:4d}   {e:10.0f}   {wt:12.0f}   {o:15.0f}   {e + wt + o:7.0f}")
e, wt, o = [(f[:, :, i][ok] / tiles[ok]).mean() for i in range(3)]
print(f" all   {e:10.0f}   {wt:12.0f}   {o:15.0f}   {e + wt + o:7.0f}   wait share {wt / (e + wt + o):.3f}")
tot = f[:, :, 4][ok]
steady = (f[:, :, 0] + f[:, :, 1] + f[:, :, 2])[ok]
rt = f[:, :, 5][ok]
print(f"whole kernel per wave: {tot.mean():.0f} cycles, of which the stamped steady loop {steady.mean():.0f} ({steady.mean() / tot.mean():.3f}); "
      f"outside it {tot.mean() - steady.mean():.0f} cycles per workgroup; shader clock {tot.mean() / rt.mean() * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz)")
names = ["pass start -> first tiles visible", "fill iteration", "unrolled steady loop", "leftover unmasked tiles", "masked tiles",
         "drain + staging-only tiles", "fallback check", "epilogue (+ next prologue issue)",
         "  entry: parameters, decode, descriptors", "  Q loads issued", "  offsets, addresses, accumulator init", "  prologue DMAs issued"]
if a.dtype == "fp8":
    names = ["pass start -> first tiles visible", "reference pre-pass (first block)", "fill block + steady pairs", "odd block out, masked blocks",
             "epilogue: normalise, store, pass-end barrier", "drain + staging-only tiles (+ next Q loads)", "exact fallback (where taken)", "next pass's prologue issue (staged: none)",
             "  entry: parameters, decode, descriptors", "  Q loads issued", "  offsets, addresses, accumulator init", "  prologue DMAs issued"]
print("(the first phase below is what remains after the four indented entry sub-phases: the wait for Q / K(0) and the barrier)")
print("phases, cycles per workgroup (sum over its passes), mean over waves [older waves 0-3 | younger 4-7]:")
ph = f[:, :, 8:20].copy()
ph[:, :, 2] += f[:, :, 0] + f[:, :, 1] + f[:, :, 2]        # the unrolled loop's cycles are kept in the segment sums
for i, nme in enumerate(names):
    print(f"  {nme:36s} {ph[:, :, i][ok].mean():9.0f}   [{ph[:, :4, i].mean():9.0f} | {ph[:, 4:, i].mean():9.0f}]")

# dispatch gap: wall time of one launch against the in-kernel lifetimes of the workgroups a CU runs one after the other
n_cu = 256
rounds = -(-n_wg // n_cu)
life_us = (f[:, :, 5].max(axis=1) / 100.0)           # s_memrealtime ticks are 10 ns
print(f"launch wall {wall_us:.1f} us; {rounds} rounds of workgroups per CU; mean workgroup lifetime {life_us.mean():.1f} us "
      f"(x rounds = {life_us.mean() * n_wg / n_cu:.1f} us): outside any workgroup {wall_us - life_us.mean() * n_wg / n_cu:.1f} us per launch "
      f"= {(wall_us - life_us.mean() * n_wg / n_cu) / max(1, n_wg / n_cu) * 1e3 * tot.mean() / rt.mean() / 10:.0f} cycles per workgroup slot")
# by dispatch round (blockIdx / 256): does the start-up phase cost more when every CU starts a workgroup at once?
print("by dispatch round (blockIdx // 256): whole workgroup | pass start -> first tiles visible | Q loads issued | prologue DMAs issued   [cycles, mean over waves]")
for r in range(rounds):
    sl = slice(r * n_cu, min(n_wg, (r + 1) * n_cu))
    print(f"  round {r}: {f[sl, :, 4].mean():9.0f} | {ph[sl, :, 0].mean():8.0f} | {ph[sl, :, 9].mean():8.0f} | {ph[sl, :, 11].mean():8.0f}")
# per wave (causal: wave w < 4 owns row block w, wave w >= 4 row block 11 - w): the phases of a pass
print("per wave, phases 0-7 in the order listed above (cycles per workgroup):")
for w in range(8):
    print(f"  wave {w}: " + " | ".join(f"{ph[:, w, i].mean():8.0f}" for i in range(8)))

# ---- per-CU timelines (absolute s_memrealtime starts, XCC_ID / HW_ID of each workgroup: d[6], d[7] of the stamp record)
if f[:, 0, 6].max() > 0:
    start = buf[:, :, 6].min(axis=1).astype(np.int64)                 # ticks of 10 ns
    end = (buf[:, :, 6] + buf[:, :, 5]).max(axis=1).astype(np.int64)
    ids = buf[:, 0, 7]
    xcc = (ids >> np.uint64(32)).astype(np.int64) & 0xF
    hw = ids.astype(np.int64) & 0xFFFFFFFF
    cu_key = xcc * (1 << 16) + ((hw >> 8) & 0xF) + (((hw >> 12) & 0x1) << 4) + (((hw >> 13) & 0x7) << 5)     # (xcc, se, sh, cu)
    t0 = start.min()
    gaps, firsts, lasts, per_cu = [], [], [], []
    for key in np.unique(cu_key):
        m = np.where(cu_key == key)[0]
        o_ = m[np.argsort(start[m])]
        per_cu.append(len(o_))
        firsts.append(start[o_[0]] - t0)
        lasts.append(end[o_].max() - t0)
        gaps += list(start[o_[1:]] - end[o_[:-1]])
    gaps = np.array(gaps, dtype=np.float64) / 100.0                   # us
    lasts = np.array(lasts, dtype=np.float64) / 100.0
    firsts = np.array(firsts, dtype=np.float64) / 100.0
    span = (end.max() - t0) / 100.0
    print(f"per-CU timelines of the LAST launch: {len(per_cu)} CUs seen, workgroups per CU min / median / max {min(per_cu)} / {int(np.median(per_cu))} / {max(per_cu)}")
    print(f"  first workgroup start -> last workgroup end: {span:.1f} us (event-timed launch: {wall_us:.1f} us: {wall_us - span:.1f} us outside any workgroup of the launch)")
    print(f"  first start of a CU after the launch's first: median {np.median(firsts):.2f} us, max {firsts.max():.2f} us")
    print(f"  gap between consecutive workgroups of a CU: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}, p90 {np.quantile(gaps, 0.9):.2f}, max {gaps.max():.2f}  (x {len(gaps) / len(per_cu):.1f} per CU = {gaps.sum() / len(per_cu):.1f} us per CU)")
    print(f"  last end of a CU before the launch's last end: median {np.median(span - lasts):.1f} us, mean {(span - lasts).mean():.1f}, max {(span - lasts).max():.1f}  (the tail)")
    for x in np.unique(xcc):
        m = xcc == x
        print(f"    XCC {x}: {int(m.sum())} workgroups that did work, busy {((end - start)[m].sum() / 100.0 / 32):.1f} us per CU, last end {(end[m].max() - t0) / 100.0:.1f} us, "
              f"workgroups started after 85 % of the launch: {int((start[m] - t0 > 0.85 * (end.max() - t0)).sum())}")
    life = (end - start) / 100.0
    print(f"  workgroup lifetime: mean {life.mean():.1f} us, p10 {np.quantile(life, 0.1):.1f}, p90 {np.quantile(life, 0.9):.1f}; by XCC: " +
          " ".join(f"{life[xcc == x].mean():.1f}" for x in np.unique(xcc)))
