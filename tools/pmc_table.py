"""profiles/r<N>_pmc_table.txt from the per-workload summaries of tools/pmc.sh (MFMA pipe busy, HBM bytes against the algorithmic
bytes, L2 hit rate, the wave-cycle split).  usage: python tools/pmc_table.py r3"""
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
names = {'cfg3_causal': ('cfg3 causal (8,32,4096,128) bf16', 4 * 8 * 32 * 4096 * 128 * 2 + 8 * 32 * 4096 * 4),
         'cfg3_noncausal': ('cfg3 non-causal', 4 * 8 * 32 * 4096 * 128 * 2 + 8 * 32 * 4096 * 4),
         'cfg4': ('cfg4 (1,16,16384,128) bf16 causal', 4 * 16 * 16384 * 128 * 2 + 16 * 16384 * 4),
         'cfg5_fp8': ('cfg5 shard (8,32,4096,128) fp8 non-causal', 3 * 8 * 32 * 4096 * 128 + 8 * 32 * 4096 * 128 * 2),
         'cfg2': ('cfg2 (4,8,1024,64) bf16 non-causal', 4 * 4 * 8 * 1024 * 64 * 2 + 4 * 8 * 1024 * 4)}
out = []
for key, (label, bytes_) in names.items():
    c = {}
    for line in open(f'profiles/{tag}_pmc_{key}_summary.txt'):
        m = re.match(r'(\S.*?)\s+(\S+)\s+mean=([0-9.e+]+)', line)
        if m:
            c.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    k = max(c, key=lambda n: c[n].get('SQ_WAVE_CYCLES', 0))
    d = c[k]
    gui = d['GRBM_GUI_ACTIVE'] / 8
    busy = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024)
    fetch, wr = d['FETCH_SIZE'] * 2 * 1024, d['WRITE_SIZE'] * 1024
    hit = d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
    wc = d['SQ_WAVE_CYCLES']
    out.append(f"{label}\n  kernel {k}\n  GRBM_GUI_ACTIVE/8 = {gui:.4g} cycles per launch; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (that x 1024 SIMDs) = {busy:.3f}\n"
               f"  HBM: FETCH_SIZE x 2 = {fetch / 1e6:.1f} MB + WRITE_SIZE = {wr / 1e6:.1f} MB = {(fetch + wr) / 1e6:.1f} MB per launch; algorithmic (Q+K+V+O+LSE) {bytes_ / 1e6:.1f} MB -> {(fetch + wr) / bytes_:.2f} x; L2 hit {hit:.3f}\n"
               f"  wave cycles (quad-cycles): issuing {d['SQ_ACTIVE_INST_ANY'] / wc:.3f} | issue-stalled {d['SQ_WAIT_INST_ANY'] / wc:.3f} | parked (waitcnt / barrier) {d['SQ_WAIT_ANY'] / wc:.3f}; "
               f"LDS bank conflicts {d.get('SQ_LDS_BANK_CONFLICT', 0):.0f}; MFMA+VALU co-execution {d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / max(1, d['SQ_VALU_MFMA_BUSY_CYCLES']):.3f} of MFMA busy\n")
# the backward step (forward + pre-pass + dK/dV kernel with the dS stores + dQ GEMM), every kernel of the summary
try:
    c = {}
    for line in open(f'profiles/{tag}_pmc_bwd_cfg3_causal_summary.txt'):
        m = re.match(r'(\S.*?)\s+(\S+)\s+mean=([0-9.e+]+)', line)
        if m:
            c.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    rows, tot = [], 0.0
    for k, d in sorted(c.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        if 'FETCH_SIZE' not in d or 'GRBM_GUI_ACTIVE' not in d:
            continue
        gui = d['GRBM_GUI_ACTIVE'] / 8
        fetch, wr = d['FETCH_SIZE'] * 2 * 1024, d['WRITE_SIZE'] * 1024
        tot += fetch + wr
        hit = d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
        rows.append(f"  {k}\n    cycles {gui:.4g}; MFMA pipe busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}; executed {d.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512 / 1e12:.3f} TFLOP; "
                    f"HBM read {fetch / 1e9:.3f} GB + written {wr / 1e9:.3f} GB; L2 hit {hit:.2f}; LDS bank conflicts {d.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
    out.append("cfg3 causal forward + backward with the dS hand-off (8,32,4096,128) bf16: per kernel and launch\n" + "\n".join(rows) +
               f"\n  HBM bytes of the whole step {tot / 1e9:.2f} GB (recompute backward: 2.6 GB): 4.33 GB of dS written by the dK/dV kernel and read once by the dQ GEMM\n")
except FileNotFoundError:
    pass
txt = (f"# {tag}: PMC passes of the final kernels (tools/campaign_prof.sh -> tools/pmc.sh, five separate rocprofv3 --pmc runs per workload; summaries {tag}_pmc_*_summary.txt)\n"
       "# FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM).  Profiled passes run at a lower clock than unprofiled ones.\n"
       "# (cfg3 causal's FETCH_SIZE moves by a few % between campaigns of the same sources: 1487 - 1558 MB per launch in three of them, L2 hit 0.72 - 0.73.)\n\n" + "\n".join(out))
open(f'profiles/{tag}_pmc_table.txt', 'w').write(txt)
print(txt)
