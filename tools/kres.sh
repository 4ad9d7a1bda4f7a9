#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra hipcc flags]  -- per-kernel register / spill / scratch summary of a device-only compile
src=$1; shift
out=/root/repo/build/asm/$(basename $src .hip).s
mkdir -p /root/repo/build/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result --cuda-device-only -S "$src" -o "$out" "$@" 2>&1 | grep -v "argument unused" 
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    n, priv, sg, sgs, vg, vgs = m.groups()
    if n.startswith("_Z"):
        flag = "  <-- SPILL" if int(vgs) or int(priv) or int(sgs) else ""
        print(f"{n[:95]:95s} vgpr {vg:>3} spill {vgs:>2} scratch {priv:>3}  sgpr {sg:>3} spill {sgs:>2}{flag}")
PY
