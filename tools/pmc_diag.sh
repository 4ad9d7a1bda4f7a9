#!/bin/bash
# usage: tools/pmc_diag.sh <outdir> <lib or ''> <prof_run args...>  -- one diagnostic rocprofv3 --pmc pass: LDS / VMEM in-flight levels
# (average latency = LEVEL / INSTS) and the LDS / texture-addresser FIFO-full stalls
out=$1; lib=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export FA_MI355_LIB=$lib
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL \
    --output-format csv -d $out/p1 -- python tools/prof_run.py "$@" > $out/p1.log 2>&1 || echo "pass failed"
python tools/pmc_summary.py $out
