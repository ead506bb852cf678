#!/bin/bash
# instruction-cache behaviour of every kernel of a workload (one --pmc run, no trace domain): fetches, hits, misses and the share of
# wave cycles spent waiting for an instruction.  usage (GPU box): bash tools/pmc_icache.sh [kernel_times.py args]
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 2
OUT=gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 tools/kernel_times.py "$@" > $OUT/log.txt 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob('gpurun_out/pmc_icache/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:34]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0))[:8]:
    a = acc[k]; c = max(n[k], 1)
    print("%-34s launches %3d  per launch: icache req %9.0f hits %9.0f misses %8.0f (+dup %8.0f) = %.1f%% miss; ifetch %9.0f; wait_inst %.1f%% of wave cycles" % (
        k, c, a['SQC_ICACHE_REQ'] / c, a['SQC_ICACHE_HITS'] / c, a['SQC_ICACHE_MISSES'] / c, a['SQC_ICACHE_MISSES_DUPLICATE'] / c,
        100 * (a['SQC_ICACHE_MISSES'] + a['SQC_ICACHE_MISSES_DUPLICATE']) / max(a['SQC_ICACHE_REQ'], 1), a['SQ_IFETCH'] / c, 100 * a['SQ_WAIT_INST_ANY'] / max(a['SQ_WAVE_CYCLES'], 1)))
PY
