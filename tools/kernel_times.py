#!/usr/bin/env python3
"""Per-kernel HIP-event times of one pass (GenPath + getPath) and the hipGraph replay time.
usage: python tools/kernel_times.py [--lib libppp_hip_x.so] [--range begin:end] config [config ...]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth  # noqa: E402

args = sys.argv[1:]
loose = False
while args and args[0] in ("--lib", "--loose"):
    if args[0] == "--lib":
        engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), args[1])
        args = args[2:]
    else:                                  # diagnostic builds whose results are wrong on purpose (tools/phase_costs.sh): errors are reported, not raised
        loose = True
        args = args[1:]
rng_kw = {}
if args and args[0] == "--range":          # one slice-range handle: --range begin:end
    b, e_ = args[1].split(":")
    rng_kw = dict(slice_begin=int(b), slice_end=int(e_))
    args = args[2:]
for name in args or ["cfg2_1m_s256"]:
    pts, cfg = synth.make_config(name)
    e = engine.Engine(0, tool_radius=cfg["tool_radius"], **rng_kw)
    e.set_cloud(pts)
    if loose:
        _sync = e.sync
        def _loose_sync():
            try:
                _sync()
            except engine.PPPError as ex:
                if not getattr(e, "_told", False):
                    print("   (loose) sync reports:", ex); e._told = True
        e.sync = _loose_sync
        e.waypoints = lambda: __import__("numpy").zeros(0)
    e.run_async(); e.sync()
    W = e.num_waypoints()
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(20):
            e.run_async()
        e.sync()
        ts.append((time.perf_counter() - t) / 20)
    e.enable_timing(True)
    acc = {}
    for _ in range(10):
        e.gen_path_async(); e.get_path_async(); e.sync()
        kt, kl = e.kernel_times(with_launches=True)
        for k, v in kt.items():
            acc[k] = acc.get(k, 0.0) + v / 10
    print("%s %s: W %d, graph replay %.4f ms, list md5 %s" % (os.path.basename(engine.LIB_PATH), name, W, min(ts) * 1e3,
                                                             "-" if rng_kw else hashlib.md5(e.waypoints().tobytes()).hexdigest()[:8]))
    print("   " + "  ".join("%s %.1f(x%d)" % (k, v * 1e3, kl[k]) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])) + "  [us]")
