#!/bin/bash
# usage: tools/pmc.sh <outdir> <prof_run args...>   -- separate rocprofv3 --pmc passes (no tracing domains)
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES"
P3="FETCH_SIZE TCC_HIT_sum"
P4="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum"
P5="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VALU_TRANS_F32"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $out/p$i -- python tools/prof_run.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $out
