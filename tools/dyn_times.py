#!/usr/bin/env python3
"""Per-kernel HIP-event times of GenPath with Dynamic_adjustment = true, and the wall time of the whole pass.
usage: python tools/dyn_times.py [--lib libppp_hip_x.so] [--walk W] config [config ...]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth  # noqa: E402

args = sys.argv[1:]
kw = {}
if args and args[0] == "--lib":
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), args[1])
    args = args[2:]
if args and args[0] == "--walk":
    kw["walk"] = int(args[1])
    args = args[2:]
for name in args or ["cfg2_1m_s256"]:
    pts, cfg = synth.make_config(name)
    e = engine.Engine(0, tool_radius=cfg["tool_radius"], dynamic_adjustment=1, **kw)
    e.set_cloud(pts)
    e.run_async(); e.sync()
    W = e.num_waypoints()
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        e.run_async()
        e.sync()
        ts.append(time.perf_counter() - t)
    e.enable_timing(True)
    e.gen_path_async(); e.get_path_async(); e.sync()
    kt, kl = e.kernel_times(with_launches=True)
    nodes = b"".join(e.nodes(s)[0].tobytes() + e.nodes(s)[1].tobytes() for s in range(e.num_slices()))
    print("%s %s: W %d, pass %.3f ms (best of 5), knots md5 %s, list md5 %s" % (
        os.path.basename(engine.LIB_PATH), name, W, min(ts) * 1e3, hashlib.md5(nodes).hexdigest()[:8],
        hashlib.md5(e.waypoints().tobytes()).hexdigest()[:8]))
    print("   " + "  ".join("%s %.0f(x%d)" % (k, v * 1e3, kl[k]) for k, v in sorted(kt.items(), key=lambda kv: -kv[1])) + "  [us]")
