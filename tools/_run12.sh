python bench.py --steps 20 --warmup 5 > gpurun_out/r2l_bench.json 2> gpurun_out/r2l_bench.err; cat gpurun_out/r2l_bench.json | cut -c1-3000; tail -2 gpurun_out/r2l_bench.err
python bench.py --config cfg3_250k_s128 --batch 64 --steps 20 --warmup 3 > gpurun_out/r2l_bench_cfg3b64.json 2> gpurun_out/r2l_bench_cfg3b64.err; cut -c1-600 gpurun_out/r2l_bench_cfg3b64.json
PPP_BENCH_FORCE_DIST=1 python bench.py --config cfg4_2m_s256 --steps 20 --warmup 5 --no-cpu-baseline --rotate 0 > gpurun_out/r2l_bench_cfg4_dist1.json 2> gpurun_out/r2l_bench_cfg4_dist1.err; cut -c1-500 gpurun_out/r2l_bench_cfg4_dist1.json; tail -2 gpurun_out/r2l_bench_cfg4_dist1.err
python tools/kernel_times.py cfg4_2m_s256 > gpurun_out/r2l_kt.log 2>&1; cat gpurun_out/r2l_kt.log
python tools/batch_check.py libppp_hip.so cfg2_1m_s256 8 > gpurun_out/r2l_batch.log 2>&1; cat gpurun_out/r2l_batch.log
python tools/slices_check.py cfg5_10m_s1024 8 > gpurun_out/r2l_slices.log 2>&1; cat gpurun_out/r2l_slices.log
python tests/tools/dyn_check.py > gpurun_out/r2l_dyn.log 2>&1; tail -5 gpurun_out/r2l_dyn.log
