#!/usr/bin/env python3
"""What would a slice cost if its y extent were shared by P workgroups?  A batch of P workpieces, each 1/P of a BASELINE plate in y
(same x range, same slice walk), run through the batched launches: P workgroups per slice position, without halo and without the
knot exchange -- a lower bound for a y-split slice kernel.
usage: python tools/parts_probe.py <config> <P> [lib]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from polishpathplanning_amd import engine, synth
from polishpathplanning_amd.hipbuf import DeviceBuffer
name, P = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3:
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), sys.argv[3])
base = synth.CONFIGS[name]
es, ws = [], []
for i in range(P):
    pts, cfg = synth.make_config(name, seed=100 + i, ny=max(base["ny"] // P, 40))
    e = engine.Engine(0, tool_radius=cfg["tool_radius"]); e.set_cloud(pts); e.gen_path(); ws.append(e.get_path()); es.append(e)
print(name, "P", P, "points per part", pts.shape[0], "S", es[0].num_slices(), "W", ws, "fast path", [e.fast_path() for e in es])
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
print("ms per batch:", " ".join("%.4f" % (t * 1e3) for t in ts))
es[0].enable_timing(True)
acc = {}
for _ in range(5):
    engine.run_batch_async(es, buf.ptr, offs, ws); engine.sync_batch(es)
    for k, v in es[0].kernel_times().items():
        acc[k] = acc.get(k, 0.0) + v / 5
es[0].enable_timing(False)
print("   " + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])) + "  [us]")
