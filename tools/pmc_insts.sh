#!/bin/bash
# instruction mix of every kernel of the bench workload (one --pmc run, no trace domain): VALU / SALU / LDS / VMEM instructions per wave
# and the share of lanes the VALU instructions had active.  usage (GPU box): bash tools/pmc_insts.sh [bench args]
set -o pipefail
export TMPDIR=/tmp
case " $* " in *" --gpus "*|*" --gpus="*) echo "$0: profile one rank (PPP_BENCH_FORCE_DIST=1 rehearses the exchange): a profiled process must not start the launcher"; exit 2;; esac
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
OUT=gpurun_out/pmc_insts
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dynamic --no-other-configs --rotate 0 --profile-passes 3 "$@" > $OUT/log.txt 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob('gpurun_out/pmc_insts/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:34]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_INSTS_VALU', 0))[:8]:
    a = acc[k]; w = max(a['SQ_WAVES'], 1)
    print("%-34s launches %3d waves/launch %7.0f  per wave: VALU %7.0f SALU %6.0f LDS %6.0f VMEM rd %5.0f wr %5.0f   VALU lanes active %4.1f%%" % (
        k, n[k], w / max(n[k], 1), a['SQ_INSTS_VALU'] / w, a['SQ_INSTS_SALU'] / w, a['SQ_INSTS_LDS'] / w, a['SQ_INSTS_VMEM_RD'] / w, a['SQ_INSTS_VMEM_WR'] / w,
        100 * a['SQ_THREAD_CYCLES_VALU'] / max(a['SQ_ACTIVE_INST_VALU'] * 64 * 4, 1)))
PY
