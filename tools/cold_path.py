#!/usr/bin/env python3
"""The path of a never-seen cloud that is already in device memory: ppp_set_cloud_device_async (conversion, bounds, walk, census, plan) +
the first ppp_run_async + the wait for the list, without torch in the process; splits per call and the kernels' own durations.
usage: python tools/cold_path.py [--lib libppp_hip_x.so] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from polishpathplanning_amd import engine, synth
from polishpathplanning_amd.hipbuf import DeviceBuffer, _rt
args = sys.argv[1:]
if args and args[0] == "--lib":
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), args[1]); args = args[2:]
name = args[0] if args else "cfg2_1m_s256"
base = sorted(synth.CONFIGS).index(name) + 1
e = engine.Engine(0, tool_radius=synth.CONFIGS[name]["tool_radius"])
buf = None
rows = []
for k in range(7):
    pts, cfg = synth.make_config(name, seed=base + 104729 * (k + 1))
    pts = np.ascontiguousarray(pts)
    if buf is None:
        buf = DeviceBuffer(pts.nbytes)
    rc = _rt().hipMemcpy(C.c_void_p(buf.ptr), pts.ctypes.data_as(C.c_void_p), C.c_size_t(pts.nbytes), 1)
    assert rc == 0
    _rt().hipDeviceSynchronize()
    t0 = time.perf_counter()
    e.set_cloud_device_async(buf.ptr, int(pts.shape[0]), 12)   # (buf is written again only after e.sync() below)
    t1 = time.perf_counter()
    e.run_async()
    t2 = time.perf_counter()
    e.sync()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
    print("cloud %d: set_cloud_device %.1f  run_async (enqueue) %.1f  wait %.1f  total %.1f us   W %d" % ((k,) + rows[-1] + (e.num_waypoints(),)), flush=True)
r = np.array(rows[1:])
print("%s %s: min over %d clouds: set_cloud_device %.1f, enqueue %.1f, wait %.1f, total %.1f us (median total %.1f)" % (
    os.path.basename(engine.LIB_PATH), name, len(r), r[:, 0].min(), r[:, 1].min(), r[:, 2].min(), r[:, 3].min(), float(np.median(r[:, 3]))))
