#!/bin/bash
# SQ counter pass (one --pmc run, no trace domain) over the bench workload: where the waves of every kernel spend their cycles.
# usage (GPU box): bash tools/pmc_sq.sh  -> table on stdout, raw csv under gpurun_out/pmc_sq
set -o pipefail
export TMPDIR=/tmp
case " $* " in *" --gpus "*|*" --gpus="*) echo "$0: profile one rank (PPP_BENCH_FORCE_DIST=1 rehearses the exchange): a profiled process must not start the launcher"; exit 2;; esac
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
OUT=gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dynamic --no-other-configs --rotate 0 --profile-passes 3 "$@" > $OUT/log.txt 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0))[:14]:
    a = acc[k]; c = max(n[k], 1); wc = a.get('SQ_WAVE_CYCLES', 1) or 1
    print("%-40s launches %3d waves %7.0f  wait_any %4.1f%%  wait_inst %4.1f%%  active %4.1f%% (lds %4.1f%%)  lds_conflict/idx_active %4.1f%%" % (
        k, c, a['SQ_WAVES']/c, 100*a['SQ_WAIT_ANY']/wc, 100*a['SQ_WAIT_INST_ANY']/wc, 100*a['SQ_ACTIVE_INST_ANY']/wc, 100*a['SQ_ACTIVE_INST_LDS']/wc,
        100*a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_LDS_IDX_ACTIVE'],1)))
PY
