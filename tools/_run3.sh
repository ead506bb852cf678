timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r2c_pytest.log
for v in "" _a _b _c _d; do
  python tools/kernel_times.py --lib libppp_hip$v.so cfg2_1m_s256 cfg5_10m_s1024 >> gpurun_out/r2c_var.log 2>&1
  python tools/batch_check.py libppp_hip$v.so cfg3_250k_s128 64 >> gpurun_out/r2c_var.log 2>&1
done
cat gpurun_out/r2c_var.log
