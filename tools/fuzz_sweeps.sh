#!/bin/bash
# The randomised GPU-vs-oracle sweeps behind profiles/<tag>_fuzz_summary.txt (run on the GPU box): full logs under gpurun_out/fuzz_<tag>/.
# usage: bash tools/fuzz_sweeps.sh <tag> [scale] [seed offset]   (scale 1: ~14 min)
TAG=${1:-r04}; K=${2:-1}; SO=${3:-0}
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
O=gpurun_out/fuzz_$TAG; mkdir -p $O
run() { # name, env, cases, seed, time limit
  echo "\$ $2 python tests/tools/fuzz_parity.py $3 $4" > $O/$1.txt
  env $2 timeout -k 10 $5 python tests/tools/fuzz_parity.py $3 $4 >> $O/$1.txt 2>&1
  tail -1 $O/$1.txt
}
run std1 PPP_X=0 $((6000 * K)) $((20261201 + SO)) $((420 * K))
run std2 PPP_X=0 $((3000 * K)) $((4242 + SO)) $((220 * K))
run odd PPP_FUZZ_ODD=1 $((1200 * K)) $((277 + SO)) $((200 * K))
run pre PPP_FUZZ_PRE=1 $((400 * K)) $((278 + SO)) $((400 * K))
run big PPP_FUZZ_BIG=1 $((60 * K)) $((279 + SO)) $((300 * K))
run tiny PPP_FUZZ_TINY=1 $((3000 * K)) $((280 + SO)) $((120 * K))
for f in std1 std2 odd pre big tiny; do head -1 $O/$f.txt; python tools/fuzz_summary.py $O/$f.txt; grep "^FAIL" $O/$f.txt | head -5; echo; done > $O/summary.txt
