#!/bin/bash
# usage: tools/pmc1.sh <outdir> <lib.so or ''> <prof_run args...>  -- ONE rocprofv3 --pmc pass (MFMA busy, wave-cycle split, clock)
# of tools/prof_run.py on the given build of the library (FA_MI355_LIB), summary appended to <outdir>/summary.txt
out=$1; lib=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export FA_MI355_LIB=$lib
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/p1 -- python tools/prof_run.py "$@" > $out/p1.log 2>&1 || echo "pass failed"
python tools/pmc_summary.py $out
python - $out <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
# kernel durations from the kernel trace of the same pass
for f in glob.glob(os.path.join(out, "p1", "**", "*kernel_trace.csv"), recursive=True):
    d = {}
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "fa::" in n:
            d.setdefault(n[:60], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for n, v in d.items():
        print(f"{n:60s} launches {len(v)} mean {sum(v)/len(v):.4f} ms min {min(v):.4f} ms")
PY
