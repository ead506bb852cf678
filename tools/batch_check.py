#!/usr/bin/env python3
"""Throughput of the batched form (ppp_run_batch_async: one launch per stage over all members) without torch in the
process, several times over to show the spread, then the per-stage kernel times of the batched launches.
usage: python tools/batch_check.py <lib, e.g. libppp_hip.so> <config> <workpieces>"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from polishpathplanning_amd import engine, synth
from polishpathplanning_amd.hipbuf import DeviceBuffer
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), sys.argv[1])
name, nb = sys.argv[2], int(sys.argv[3])
es, ws = [], []
rng = np.random.default_rng(7)
for i in range(nb):
    pts, cfg = synth.make_config(name, seed=100 + i, amp=float(synth.CONFIGS[name]["amp"] * rng.uniform(0.5, 1.5)))
    e = engine.Engine(0, tool_radius=cfg["tool_radius"]); e.set_cloud(pts); e.gen_path(); ws.append(e.get_path()); es.append(e)
offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
buf = DeviceBuffer(sum(ws) * 24)
for _ in range(3):
    engine.run_batch_async(es, buf.ptr, offs, ws); engine.sync_batch(es)
ts = []
for rep in range(5):
    t = time.perf_counter()
    for _ in range(20):
        engine.run_batch_async(es, buf.ptr, offs, ws)
    engine.sync_batch(es)
    ts.append((time.perf_counter() - t) / 20)
print(sys.argv[1], name, "x", nb, "ms per batch:", " ".join("%.3f" % (t * 1e3) for t in ts), "-> %.3e wp/s" % (sum(ws) / min(ts)))
es[0].enable_timing(True)
acc = {}
for _ in range(5):
    engine.run_batch_async(es, buf.ptr, offs, ws); engine.sync_batch(es)
    for k, v in es[0].kernel_times().items():
        acc[k] = acc.get(k, 0.0) + v / 5
es[0].enable_timing(False)
print("   " + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])) + "  [us]")
