python bench.py --steps 20 --warmup 5 > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; cat gpurun_out/r2h_bench.json; tail -3 gpurun_out/r2h_bench.err
PPP_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate 0 > gpurun_out/r2h_bench_dist1.json 2> gpurun_out/r2h_bench_dist1.err; cut -c1-400 gpurun_out/r2h_bench_dist1.json; tail -3 gpurun_out/r2h_bench_dist1.err
python tools/kernel_times.py --range 384:512 cfg5_10m_s1024 > gpurun_out/r2h_kt_range.log 2>&1; cat gpurun_out/r2h_kt_range.log
