timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2i_pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r2i_pytest.log
python tools/kernel_times.py cfg5_10m_s1024 > gpurun_out/r2i_kt.log 2>&1; python tools/kernel_times.py --range 384:512 cfg5_10m_s1024 >> gpurun_out/r2i_kt.log 2>&1; cat gpurun_out/r2i_kt.log
python tools/slices_check.py cfg5_10m_s1024 8 > gpurun_out/r2i_slices.log 2>&1; cat gpurun_out/r2i_slices.log
