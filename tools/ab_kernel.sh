#!/bin/bash
# A/B of one kernel between two builds of the library on the SAME box under rocprofv3 (event timers of two sessions differ by more
# than the effects that are left to measure).  usage (GPU box): bash tools/ab_kernel.sh <kernel substring> <libA.so> <libB.so> [config ...]
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
PAT=$1; LA=$2; LB=$3; shift 3
CFGS=${@:-cfg2_1m_s256}
OUT=gpurun_out/ab_kernel; rm -rf $OUT; mkdir -p $OUT
for cfg in $CFGS; do
  for rep in 1 2; do
    for L in $LA $LB; do
      if [[ $cfg == *:b* ]]; then c=${cfg%%:b*}; nb=${cfg##*:b}; CMD="tools/batch_check.py $L $c $nb"; else CMD="tools/kernel_times.py --lib $L $cfg"; fi
      rocprofv3 --kernel-trace --output-format csv -d $OUT/${cfg//:/_}_${L}_$rep -- python3 $CMD > $OUT/${cfg//:/_}_${L}_$rep.log 2>&1 || echo "failed: $cfg $L"
    done
  done
done
python3 - "$OUT" "$PAT" "$LA" "$LB" $CFGS <<'PY'
import csv, glob, sys, statistics
out, pat, la, lb = sys.argv[1:5]
for cfg in sys.argv[5:]:
    row = []
    for L in (la, lb):
        d = []
        for rep in (1, 2):
            for fn in glob.glob("%s/%s_%s_%d/**/*kernel_trace.csv" % (out, cfg.replace(":", "_"), L, rep), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if pat in r["Kernel_Name"]:
                        d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        d.sort()
        row.append((statistics.median(d) if d else float("nan"), len(d)))
    print("%-22s %-14s  %s median %.2f us (%d launches)   %s median %.2f us (%d launches)   B/A %.3f" % (cfg, pat, la, row[0][0], row[0][1], lb, row[1][0], row[1][1], row[1][0] / row[0][0]))
PY
