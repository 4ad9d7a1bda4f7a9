# round-end bench lines on the GPU box (driver settings, defaults, the other BASELINE workloads, forward + backward, all shapes,
# the explicit two-rank rehearsal, the RCCL path at world size 1) and the in-kernel stamps of the diagnostic build
set -x
O=${1:-gpurun_out/r4v}; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver.json 2> $O/bench_cfg3_driver.err &&
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err &&
python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err &&
python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err &&
python bench.py --workload cfg3nc --no-cpu-baseline > $O/bench_cfg3nc.json 2> $O/bench_cfg3nc.err &&
python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err &&
python bench.py --mode fwdbwd --no-cpu-baseline > $O/bench_fwdbwd.json 2> $O/bench_fwdbwd.err &&
timeout -k 10 600 python tools/report_all.py > $O/all_configs.txt 2> $O/all_configs.err &&
timeout -k 10 300 python tools/bench_bwd.py --configs cfg2,cfg3,cfg3nc,cfg3fp16,cfg4,ref-bwd,d64,ref-main > $O/bwd_bench.txt 2> $O/bwd_bench.err &&
(bash tools/build_variant.sh recompute -DFA_BWD_DS_DISABLE > /dev/null 2>&1; timeout -k 10 400 python tools/sweep_bwd_handoff.py > $O/bwd_handoff_sweep.txt 2>&1; true) &&
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --rehearse-gather --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 10 --warmup 3 --no-cpu-baseline --no-attainable > $O/bench_n1_nccl.json 2> $O/bench_n1_nccl.err &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-dist --dist-backend nccl --scaling strong --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-attainable > $O/bench_n1_nccl_strong_cfg4.json 2> $O/bench_n1_nccl_strong_cfg4.err &&
python bench.py --workload cfg5 --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_strong_world1.json 2> $O/bench_cfg5_strong_world1.err &&
timeout -k 10 400 python tools/shard_shapes.py > $O/shard_shapes.txt 2> $O/shard_shapes.err &&
timeout -k 10 400 python tools/compare_sdpa.py > $O/compare_torch_sdpa.txt 2> $O/compare_torch_sdpa.err
echo rc=$?
if [ -f build/libstamp.so ]; then
  FA_MI355_LIB=build/libstamp.so python tools/stamps.py --causal 0 > $O/stamps_cfg3.txt 2>&1
  FA_MI355_LIB=build/libstamp.so python tools/stamps.py --causal 1 >> $O/stamps_cfg3.txt 2>&1
  FA_MI355_LIB=build/libstamp.so python tools/stamps.py --dtype fp8 --causal 0 > $O/stamps_cfg5.txt 2>&1
  FA_MI355_LIB=build/libstamp.so python tools/stamps.py --dtype fp8 --causal 1 >> $O/stamps_cfg5.txt 2>&1
fi
tail -c 700 $O/bench_cfg3_driver.json; cat $O/all_configs.txt | tail -9
