#!/bin/bash
# What every phase of the per-slice kernel costs: builds that leave the kernel after phase k (make variant ... -DWIN_STOP_AFTER=k,
# built beforehand: tools/phase_costs.sh build), each run under rocprofv3 for duration and instruction counters; the difference
# of two consecutive rows is the phase's bill.  usage (GPU box): bash tools/phase_costs.sh run [config]
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
if [ "$1" = build ]; then
  for k in 0 1 2 3 4 5 6 7 8; do make -C polishpathplanning_amd/csrc variant NAME=stop$k DEFS=-DWIN_STOP_AFTER=$k > /dev/null & done
  wait; ls -la polishpathplanning_amd/libppp_hip_stop*.so; exit 0
fi
CFG=${2:-cfg2_1m_s256}
OUT=gpurun_out/phase_costs_$CFG
rm -rf $OUT; mkdir -p $OUT
for lib in stop0 stop1 stop2 stop3 stop4 stop5 stop6 stop7 stop8 full; do
  L=libppp_hip_$lib.so; [ $lib = full ] && L=libppp_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$lib -- python3 tools/kernel_times.py --lib $L --loose $CFG > $OUT/t_$lib.log 2>&1 || echo "trace $lib failed"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p_$lib -- python3 tools/kernel_times.py --lib $L --loose $CFG > $OUT/p_$lib.log 2>&1 || echo "pmc $lib failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
prev = None
print("%-6s %9s %9s %9s %9s %9s %8s | phase: %8s %8s %8s" % ("build", "us", "VALU/wv", "SALU/wv", "LDS/wv", "waves", "wait%", "us", "VALU", "LDS"))
for lib in ["stop%d" % k for k in range(9)] + ["full"]:
    dur = []
    for fn in glob.glob("%s/t_%s/**/*kernel_trace.csv" % (out, lib), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Kernel_Name"].startswith("void k_win_slice") or r["Kernel_Name"].startswith("k_win_slice"):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    acc = collections.defaultdict(float)
    for fn in glob.glob("%s/p_%s/**/*counter_collection.csv" % (out, lib), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_win_slice" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    w = max(acc["SQ_WAVES"], 1.0)
    dur.sort()
    d = dur[len(dur) // 2] if dur else float("nan")
    row = (d, acc["SQ_INSTS_VALU"] / w, acc["SQ_INSTS_SALU"] / w, acc["SQ_INSTS_LDS"] / w)
    ph = (row[0] - prev[0], row[1] - prev[1], row[3] - prev[3]) if prev else (row[0], row[1], row[3])
    print("%-6s %9.2f %9.0f %9.0f %9.0f %9.0f %8.1f | %15.2f %8.0f %8.0f" % (lib, row[0], row[1], row[2], row[3], w, 100 * acc["SQ_WAIT_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1), ph[0], ph[1], ph[2]))
    prev = row
PY
