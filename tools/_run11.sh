timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2k_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r2k_pytest.log
bash tools/collect_profiles.sh r02b_cfg2 > gpurun_out/r2k_prof_cfg2.log 2>&1; tail -3 gpurun_out/r2k_prof_cfg2.log
PPP_PROFILE_WORKLOAD=cfg3_250k_s128_b64 bash tools/collect_profiles.sh r02b_cfg3b64 --config cfg3_250k_s128 --batch 64 > gpurun_out/r2k_prof_cfg3.log 2>&1; tail -3 gpurun_out/r2k_prof_cfg3.log
PPP_PROFILE_WORKLOAD=cfg5_10m_s1024 bash tools/collect_profiles.sh r02b_cfg5 --config cfg5_10m_s1024 > gpurun_out/r2k_prof_cfg5.log 2>&1; tail -3 gpurun_out/r2k_prof_cfg5.log
ls gpurun_out/prof_r02b_*/summary/
