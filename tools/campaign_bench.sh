set -x
O=gpurun_out/r2v; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver.json 2> $O/bench_cfg3_driver.err &&
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err &&
python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err &&
python bench.py --mode fwdbwd --no-cpu-baseline > $O/bench_fwdbwd.json 2> $O/bench_fwdbwd.err &&
timeout -k 10 600 python tools/report_all.py > $O/all_configs.txt 2> $O/all_configs.err &&
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
echo rc=$?
tail -c 600 $O/bench_cfg3_driver.json; tail -c 400 $O/bench_cfg5.json; cat $O/all_configs.txt | tail -9
