timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2e_pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r2e_pytest.log
bash tools/collect_profiles.sh r02a_cfg2 > gpurun_out/r2e_prof_cfg2.log 2>&1; tail -30 gpurun_out/r2e_prof_cfg2.log
