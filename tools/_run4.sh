timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r2d_pytest.log
for c in cfg2_1m_s256 cfg3_250k_s128 cfg5_10m_s1024; do echo $c; python tools/stamps.py $c; done > gpurun_out/r2d_stamps.log 2>&1
cat gpurun_out/r2d_stamps.log
