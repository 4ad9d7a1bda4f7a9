"""Summarise the instruction mix of one kernel in a hipcc --save-temps .s file as runs of instruction classes
(M = MFMA, V = VALU, T = transcendental, L = LDS, G = global/buffer, S = scalar/other, W = s_waitcnt, B = barrier).
usage: isa_runs.py file.s kernel-name-substring [min-run]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))


def cls(op):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "T"
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "G"
    if op.startswith("s_waitcnt"):
        return "W"
    if op.startswith("s_barrier"):
        return "B"
    return "S"


out = []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")):
        continue
    if t.endswith(":"):
        out.append("\n" + t + "\n")
        continue
    out.append(cls(t.split()[0]))
text = "".join(out)
# compress runs
print(re.sub(r"(.)\1{3,}", lambda m: f"{m.group(1)}{{{len(m.group(0))}}}", text))
