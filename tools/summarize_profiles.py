"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the small files kept under
profiles/: <tag>_kernel_stats.csv (the --stats table) and <tag>_traffic.json (per-kernel mean
FETCH_SIZE / WRITE_SIZE per launch, KB as rocprofv3 reports them, plus corrected HBM bytes:
FETCH_SIZE is doubled as MI355X_MICROARCH.md 'HBM' prescribes for gfx950 streaming reads)."""
import collections, csv, glob, json, os, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))


def kernel_key(name):
    """'void k_slice_kd<false>(HIP_vector_type<...' -> 'k_slice_kd' (the name bench.py's event timers use; the <true>
    instantiations are the arena passes, timed as '<kernel>_arena')."""
    base = name.split("(")[0].replace("void ", "").strip()
    if "<" in base:
        base, targ = base.split("<", 1)
        if targ.startswith("true") and base in ("k_slab_sort", "k_slice_kd"):   # the LDS-overflow passes of the slab path
            base += "_arena"
    return base


def pmc(sub, name):
    """mean counter value per launch, keyed by kernel -- and by grid size where one kernel is launched with several (a batch
    split into two halves beside a full-width roofline launch must not be averaged into one figure)"""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[(kernel_key(r["Kernel_Name"]), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    grids = collections.defaultdict(set)
    for (k, g) in acc:
        grids[k].add(g)
    vals, cnts = {}, {}
    for (k, g), v in acc.items():
        key = k if len(grids[k]) == 1 else "%s@grid%s" % (k, g)
        vals[key] = sum(v) / len(v); cnts[key] = len(v)
    return vals, cnts


fetch, nf = pmc("pmc_fetch", "FETCH_SIZE")
write, nw = pmc("pmc_write", "WRITE_SIZE")
table = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    table[k] = {"fetch_size_kb": f, "write_size_kb": w, "launches_sampled": nf.get(k, nw.get(k, 0)),
                "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump({"tag": tag, "workload": os.environ.get("PPP_PROFILE_WORKLOAD", "cfg2_1m_s256"), "units": "rocprofv3 FETCH_SIZE/WRITE_SIZE in KB, mean per launch; hbm_bytes = (2*FETCH + WRITE)*1024",
           "kernels": table}, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
print(open(os.path.join(dst, "%s_traffic.json" % tag)).read()[:3000])
if stats:
    print(open(stats[0]).read()[:3000])
