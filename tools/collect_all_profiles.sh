#!/bin/bash
# Run on the GPU box (via gpurun): the profiles of every BASELINE configuration on the current build -- rocprofv3 kernel
# trace + the two PMC traffic passes each (tools/collect_profiles.sh), and the SQ counter pass for cfg 2 and the cfg 3 batch.
# Usage: bash tools/collect_all_profiles.sh r03f      -> gpurun_out/prof_<tag>_<workload>/summary/*
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { # workload key, bench args...
  local key=$1; shift
  PPP_PROFILE_WORKLOAD=$key bash tools/collect_profiles.sh ${TAG}_$key "$@" > gpurun_out/prof_${TAG}_$key.log 2>&1
  echo "== $key"; tail -4 gpurun_out/prof_${TAG}_$key.log | cut -c1-200
}
run cfg2_1m_s256
run cfg3_250k_s128_b64 --config cfg3_250k_s128 --batch 64
run cfg4_2m_s256 --config cfg4_2m_s256
run cfg5_10m_s1024 --config cfg5_10m_s1024
run cfg2_1m_s256_dyn --dynamic
bash tools/pmc_sq.sh > gpurun_out/prof_${TAG}_cfg2_sq_counters.txt 2>&1
bash tools/pmc_sq.sh --config cfg3_250k_s128 --batch 64 > gpurun_out/prof_${TAG}_cfg3b64_sq_counters.txt 2>&1
bash tools/pmc_sq.sh --config cfg5_10m_s1024 > gpurun_out/prof_${TAG}_cfg5_sq_counters.txt 2>&1
