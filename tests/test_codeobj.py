"""Build-time guard on the shipped code object: no kernel of the default path may spill vector registers or own a
private (scratch) segment.  A scratch reload sits behind an `s_waitcnt vmcnt(0)` that hipcc places itself, and that wait
drains the hand-counted LDS-DMA ring of the attention kernels -- a silent performance loss that the
parity tests cannot see.  Reads the gfx950 code object out of the in-tree library with the ROCm LLVM tools (CPU only)."""
import os
import re
import shutil
import subprocess

import pytest

from flash_attention_impls_amd import _build

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes(tmp_path):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM tools not installed")
    lib = shutil.copy(_build.build(), tmp_path / "lib.so")           # (the tool extracts next to its input)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert objs, "no gfx950 code object found in the library"
    kernels = {}
    for f in objs:
        txt = subprocess.run([readelf, "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            kernels[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                             for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return kernels


def test_no_kernel_spills_or_uses_scratch(tmp_path):
    kernels = _kernel_notes(tmp_path)
    attn = {n: k for n, k in kernels.items() if "fa_fwd" in n or "fa_bwd" in n}
    assert len(attn) >= 20, sorted(kernels)                            # forward (16-bit, fp8) and the three backward kernels
    bad = {n: k for n, k in attn.items() if k["vgpr_spill_count"] or k["private_segment_fixed_size"]}
    assert not bad, f"kernels with vector spills / scratch: {bad}"
    # scalar spills go to lanes of a VGPR (v_writelane / v_readlane), never to memory: launch parameters parked between a
    # kernel's prologue and epilogue.  Bounded for every kernel, and absent from the headline kernels (head_dim-128 forward):
    # parked scalars there cost more than their instructions -- with ~80 of them (the persistent-workgroup experiment of round 4,
    # profiles/r4_persistent_workgroups_experiment.patch) the scheduler gave up the MFMA / LDS-read / VALU interleave of the
    # steady loop (73 instead of 12 back-to-back MFMA pairs, -4 %)
    assert all(k["sgpr_spill_count"] <= 40 for k in attn.values()), {n: k["sgpr_spill_count"] for n, k in attn.items() if k["sgpr_spill_count"]}
    head = {n: k for n, k in attn.items() if "fa_fwd_kernel16" in n and "Lb0ELi128ELi8E" in n}   # 16-bit Q/K/V, 256-row workgroups
    assert head and all(k["sgpr_spill_count"] == 0 for k in head.values()), head
    # two waves per SIMD: <= 256 registers (the 128-row form of the head_dim-128 forward, ...Li128ELi4E, runs one wave per SIMD)
    # (fa_bwd_wide_ds_kernel, head_dim 144 .. 256, is a one-wave-per-SIMD kernel like the wide forward)
    assert all(k["vgpr_count"] <= 256 for n, k in attn.items()
               if ("fa_fwd_kernel16" in n and "ELi4E" not in n) or ("fa_bwd" in n and "fa_bwd_wide" not in n))


def test_steady_loops_touch_no_parked_scalars(tmp_path):
    """The unrolled steady loop of every head_dim-128 forward kernel (the straight-line run of instructions with the most MFMAs:
    four 64-key tiles, 256 of them) holds no v_readlane / v_writelane (a scalar parked in a vector register and fetched back),
    no scratch access and no s_waitcnt vmcnt(0) (which would drain the hand-counted LDS-DMA ring), and keeps its interleave: at most
    16 of its 256 MFMAs directly follow another one (12 in the shipped build; 73 - 84 where the scheduler had given the pattern up)."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("ROCm LLVM tools not installed")
    lib = shutil.copy(_build.build(), tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    seen = 0
    for f in [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]:
        txt = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(_ZN2fa15fa_fwd_kernel16[^>]*Lb0ELi128ELi8E[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.M | re.S):
            name, body = m.group(1), m.group(2)
            # instructions with their addresses; a loop = the address range of a BACKWARD branch (target <= branch); the steady
            # loop is the one with the most MFMAs (the forward branches inside it -- the s_setprio statements' own -- do not split it)
            ins = []
            for line in body.splitlines():
                mm = re.match(r"\s+(\S.*?)\s*// ([0-9A-F]+): ", line)
                if mm:
                    ins.append((int(mm.group(2), 16), mm.group(1), line))
            base = ins[0][0]
            loops = []
            for addr, text, line in ins:
                t = re.search(r"<[^>+]*\+0x([0-9a-f]+)>\s*$", line)
                if text.startswith(("s_cbranch", "s_branch")) and t and base + int(t.group(1), 16) <= addr:
                    loops.append([x[1] for x in ins if base + int(t.group(1), 16) <= x[0] <= addr])
            assert loops, name
            big = [r for r in loops if sum("v_mfma" in x for x in r) >= 256]        # the steady loop and the loops around it
            assert big, (name, max(sum("v_mfma" in x for x in r) for r in loops))
            loop = min(big, key=len)                                                # ... the innermost of them
            assert sum("v_mfma" in x for x in loop) == 256, (name, sum("v_mfma" in x for x in loop))
            bad = [x.strip() for x in loop if re.search(r"v_readlane|v_writelane|scratch_|vmcnt\(0\)", x)]
            assert not bad, (name, bad[:5])
            ops = [x.split()[0] for x in loop if not x.startswith(("s_waitcnt", "s_nop"))]
            b2b = sum(1 for i in range(len(ops) - 1) if ops[i].startswith("v_mfma") and ops[i + 1].startswith("v_mfma"))
            assert b2b <= 16, (name, b2b)
            seen += 1
    assert seen >= 4, seen


def test_head_dim_64_forward_fits_two_workgroups_per_cu(tmp_path):
    """The head_dim-64 16x16 kernel is launched for two workgroups per CU: that needs <= 128 registers per lane."""
    kernels = _kernel_notes(tmp_path)
    d64 = {n: k for n, k in kernels.items() if "fa_fwd_kernel16" in n and "Li64E" in n}
    assert d64 and all(k["vgpr_count"] <= 128 for k in d64.values()), d64
