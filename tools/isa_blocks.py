"""Per-basic-block instruction counts of one kernel in a hipcc --save-temps .s file.
usage: isa_blocks.py file.s kernel-name-substring"""
import collections
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
KEYS = ["v_mfma", "v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov", "v_exp", "v_cvt_pk", "v_fma", "v_mul_f32", "v_sub_f32", "v_add_f32",
        "v_mov", "v_cndmask", "ds_read_b128", "ds_read_b64_tr", "buffer_load", "s_waitcnt", "s_barrier", "s_nop", "scratch_"]
blocks = []
cur = ("entry", collections.Counter(), [0])
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", "//")):
        continue
    if t.startswith(".LBB") and ":" in t:
        blocks.append(cur)
        cur = (t.split(":")[0], collections.Counter(), [0])
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    cur[2][0] += 1
    for k in KEYS:
        if op.startswith(k):
            cur[1][k] += 1
            break
    else:
        cur[1]["v_other" if op.startswith("v_") else ("s_other" if op.startswith("s_") else "other")] += 1
blocks.append(cur)
for name, cnt, n in blocks:
    if n[0] >= 40:
        print(f"{name:12s} n={n[0]:5d} " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])))
