timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2g_pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r2g_pytest.log
python tools/slices_check.py cfg5_10m_s1024 8 > gpurun_out/r2g_slices.log 2>&1; cat gpurun_out/r2g_slices.log
python bench.py --config cfg3_250k_s128 --batch 64 --steps 20 --warmup 3 > gpurun_out/r2g_bench_cfg3b64.json 2> gpurun_out/r2g_bench_cfg3b64.err; cut -c1-1500 gpurun_out/r2g_bench_cfg3b64.json; tail -3 gpurun_out/r2g_bench_cfg3b64.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; cat gpurun_out/r2g_bench.json; tail -3 gpurun_out/r2g_bench.err
