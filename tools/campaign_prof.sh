# round-end profile set (GPU box): rocprofv3 --stats of the bench commands, then the five-pass PMC runs of the kernels whose HBM
# traffic bench.py reports (round 3: + cfg4 and cfg2).  Every rocprofv3 call is either --kernel-trace --stats or --pmc, never both.
set -x
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=${1:-gpurun_out/r3w}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-power > $O/bench_cfg3_under_rocprofv3.json 2> $O/prof_cfg3.err &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --no-power > $O/bench_cfg5_under_rocprofv3.json 2> $O/prof_cfg5.err &&
bash tools/pmc.sh $O/pmc_cfg3_causal --causal 1 --iters 3 > $O/pmc_cfg3_causal_summary.txt 2>&1 &&
bash tools/pmc.sh $O/pmc_cfg3_noncausal --causal 0 --iters 3 > $O/pmc_cfg3_noncausal_summary.txt 2>&1 &&
bash tools/pmc.sh $O/pmc_cfg5 --causal 0 --dtype fp8 --iters 3 > $O/pmc_cfg5_fp8_summary.txt 2>&1 &&
bash tools/pmc.sh $O/pmc_cfg4 --B 1 --H 16 --S 16384 --causal 1 --iters 3 > $O/pmc_cfg4_summary.txt 2>&1 &&
bash tools/pmc.sh $O/pmc_cfg2 --B 4 --H 8 --S 1024 --D 64 --causal 0 --iters 20 > $O/pmc_cfg2_summary.txt 2>&1 &&
bash tools/pmc.sh $O/pmc_bwd_cfg3_causal --causal 1 --bwd 1 --iters 3 > $O/pmc_bwd_cfg3_causal_summary.txt 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwdbwd -- python3 bench.py --mode fwdbwd --steps 50 --warmup 5 --no-cpu-baseline --no-power > $O/bench_cfg3_fwdbwd_under_rocprofv3.json 2> $O/prof_fwdbwd.err
# then, in the tree: python tools/update_traffic.py <workload> <summary>  (forward workloads)
#                    python tools/update_traffic.py cfg3_fwdbwd <bwd summary> --unit=fa_capi.hip,fa_bwd_capi.hip ; python tools/pmc_table.py r<N>
echo rc=$?
find $O -name "*.csv" -size +2000k -delete
for f in $(find $O -name "*kernel_stats.csv"); do head -5 $f; done
tail -5 $O/pmc_cfg4_summary.txt
