"""Summarises rocprofv3 --pmc CSV output per kernel (mean per dispatch)."""
import csv, sys, glob, collections
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in acc.items():
            if k.startswith("k_"):
                print("  %-16s" % k, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())), "(n=%d)" % len(next(iter(cs.values()))))
