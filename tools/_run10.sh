timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2j_pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r2j_pytest.log
python tools/kernel_times.py cfg2_1m_s256 cfg3_250k_s128 cfg5_10m_s1024 > gpurun_out/r2j_kt.log 2>&1; cat gpurun_out/r2j_kt.log
python tools/batch_check.py libppp_hip.so cfg3_250k_s128 64 > gpurun_out/r2j_batch.log 2>&1; cat gpurun_out/r2j_batch.log
