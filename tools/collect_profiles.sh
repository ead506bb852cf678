#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + the HBM traffic counters of bench.py's
# workload.  --pmc passes are separate runs and carry no other trace domain (pool rule).
# Usage: tools/collect_profiles.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
case " $* " in *" --gpus "*|*" --gpus="*) echo "collect_profiles.sh: profile one rank (PPP_BENCH_FORCE_DIST=1 rehearses the exchange): a profiled process must not start the launcher"; exit 2;; esac
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# (one handle: a launch alone on the device, what roofline.avg_launch_ms reports; PPP_PROFILE_HANDLES=3 profiles the default loop,
#  consecutive steps taking turns on three handles -- roofline.avg_launch_ms_steps_taking_turns -- without the one-handle loop beside it)
H=${PPP_PROFILE_HANDLES:-1}
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-dynamic --no-other-configs --rotate 0 --profile-passes 3 --handles $H --no-single-handle $@"
if [ "$H" != "1" ]; then ARGS="--steps 200 --warmup 10 --no-cpu-baseline --no-dynamic --no-other-configs --rotate 0 --profile-passes 2 --handles $H --no-single-handle $@"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "write failed"
python3 tools/summarize_profiles.py $OUT $TAG
