#!/bin/bash
# The bench lines and rehearsals kept under profiles/ for a build (run on the GPU box): usage bash tools/final_lines.sh <tag>
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
O=gpurun_out/final_$TAG; mkdir -p $O
python bench.py > $O/${TAG}_cfg2_bench.json 2> $O/cfg2_bench.err
python bench.py --config cfg3_250k_s128 --batch 64 --no-cpu-baseline > $O/${TAG}_cfg3b64_bench.json 2>> $O/bench.err
python bench.py --config cfg4_2m_s256 --no-cpu-baseline > $O/${TAG}_cfg4_bench.json 2>> $O/bench.err
PPP_BENCH_FORCE_DIST=1 python bench.py --config cfg4_2m_s256 --no-cpu-baseline --no-dynamic --rotate 0 > $O/${TAG}_cfg4_onerank_gather_bench.json 2>> $O/bench.err
python bench.py --config cfg5_10m_s1024 --no-cpu-baseline > $O/${TAG}_cfg5_bench.json 2>> $O/bench.err
python tools/slices_check.py cfg5_10m_s1024 8 > $O/${TAG}_cfg5_slices_rehearsal.txt 2>&1
python tools/dyn_times.py cfg2_1m_s256 > $O/${TAG}_dyn_times.txt 2>&1
python tools/brute_times.py > $O/${TAG}_brute_times.txt 2>&1
python tools/cold_path.py > $O/${TAG}_cold_path.txt 2>&1
echo "lines done"; tail -1 $O/${TAG}_cold_path.txt; tail -2 $O/${TAG}_cfg5_slices_rehearsal.txt
