#!/usr/bin/env python3
"""Throughput of the batched form (ppp_run_batch_async: one hipGraph, a branch per workpiece) without torch in the
process.  usage: python tools/batch_check.py <lib, e.g. libppp_hip.so> <config> <workpieces>
The time per batch is bimodal from process to process (how the graph's branches land on the hardware queues)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from polishpathplanning_amd import engine, synth
from polishpathplanning_amd.hipbuf import DeviceBuffer
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), sys.argv[1])
name, nb = sys.argv[2], int(sys.argv[3])
es, ws = [], []
for i in range(nb):
    pts, cfg = synth.make_config(name, seed=100 + i)
    e = engine.Engine(0, tool_radius=cfg["tool_radius"]); e.set_cloud(pts); e.gen_path(); ws.append(e.get_path()); es.append(e)
offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
buf = DeviceBuffer(sum(ws) * 24)
for _ in range(3):
    engine.run_batch_async(es, buf.ptr, offs, ws); engine.sync_batch(es)
t = time.perf_counter()
for _ in range(20):
    engine.run_batch_async(es, buf.ptr, offs, ws); engine.sync_batch(es)
dt = (time.perf_counter() - t) / 20
print(sys.argv[1], name, "x", nb, "%.3f ms per batch, %.3e wp/s" % (dt * 1e3, sum(ws) / dt))
