#!/usr/bin/env python3
"""Counts of a tests/tools/fuzz_parity.py log by walk, pairing, dynamic adjustment and cloud size (the tables of profiles/*_fuzz_summary.txt).
usage: python tools/fuzz_summary.py log [log ...]"""
import collections
import re
import sys

for path in sys.argv[1:]:
    c = collections.Counter()
    ok = bad = 0
    last = ""
    for line in open(path):
        m = re.match(r"(ok  |FAIL) .*case \d+: .* walk (\d) pairing (\d) dyn (\d) .* n (\d+)", line)
        if not m:
            if re.match(r"\d+ cases, \d+ failures", line):
                last = line.strip()
            continue
        ok += m.group(1) == "ok  "
        bad += m.group(1) == "FAIL"
        c["walk %s" % m.group(2)] += 1
        c["pairing %s" % m.group(3)] += 1
        c["dyn %s" % m.group(4)] += 1
        n = int(m.group(5))
        c["points < 5k" if n < 5000 else ("points 5k-50k" if n < 50000 else ("points 50k-500k" if n < 500000 else "points >= 500k"))] += 1
        if "both fail" in line or "both sides" in line:
            c["both sides report the same failing slice"] += 1
    print(last)
    print("cases run (not skipped): %d ok, %d FAIL" % (ok, bad))
    for k in sorted(c):
        print("  %-45s %d" % (k, c[k]))
