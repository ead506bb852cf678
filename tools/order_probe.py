#!/usr/bin/env python3
"""Does the index order of the cloud matter to the pipeline's speed?  The BASELINE clouds are randomly permuted (the reference's
results depend on index order); a scanner delivers rows.  The same plate permuted, in scan rows along y (x-major),
in scan rows along x (y-major): ms per pass and the kernels' own times.  usage: python tools/order_probe.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polishpathplanning_amd import engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_1m_s256"
pts, cfg = synth.make_config(name)
orders = {"permuted": pts, "rows along y (sorted by x)": pts[np.argsort(pts[:, 0], kind="stable")],
          "rows along x (sorted by y)": pts[np.argsort(pts[:, 1], kind="stable")]}
for label, p in orders.items():
    e = engine.Engine(0, tool_radius=cfg["tool_radius"]); e.set_cloud(np.ascontiguousarray(p))
    planned = e.fast_path()
    e.run_async(); e.sync()
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(10):
            e.run_async()
        e.sync(); ts.append((time.perf_counter() - t) / 10)
    e.enable_timing(True); e.gen_path_async(); e.get_path_async(); e.sync()
    print("%-16s %-28s window path planned %-5s kept %-5s W %6d  %.4f ms  %s" % (name, label, planned, e.fast_path(), e.num_waypoints(), min(ts) * 1e3,
          " ".join("%s %.1f" % (k, v * 1e3) for k, v in sorted(e.kernel_times().items(), key=lambda kv: -kv[1]))), flush=True)
    e.close()
