#!/usr/bin/env python3
"""Slice-range sharding (SURVEY.md 8e case ii) rehearsed on ONE GPU: `world` range handles run one after
the other, their blocks are concatenated in a device buffer and finished on handle 0.  Prints the time
of the slowest range (what a rank of an N-GPU job spends before the gather), the finishing time, and
checks the list against the unsharded handle.  usage: python tools/slices_check.py [config] [world]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth  # noqa: E402
from polishpathplanning_amd.robot_path import slice_ranges  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_10m_s1024"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    pts, cfg = synth.make_config(name)
    R = cfg["tool_radius"]
    one = engine.Engine(0, tool_radius=R)
    one.set_cloud(pts)
    S = one.gen_path()
    W = one.get_path()
    one.run_async(); one.sync()   # the first pass of a plan is enqueued directly ...
    one.run_async(); one.sync()   # ... the second captures the graph
    t = time.perf_counter()
    for _ in range(5):
        one.run_async(); one.sync()
    t_one = (time.perf_counter() - t) / 5
    hip = C.CDLL("libamdhip64.so")
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), C.c_size_t(max(W, 1) * 24)) == 0
    engines, counts, off, t_ranges = [], None, 0, []
    for b, e in slice_ranges(S, world):
        if b == e:
            continue
        g = engine.Engine(0, tool_radius=R, slice_begin=b, slice_end=e)
        g.set_cloud(pts)
        g.gen_path(); g.get_path()
        g.run_async(); g.sync()
        g.run_async(); g.sync()
        t = time.perf_counter()
        for _ in range(5):
            g.run_async(); g.sync()
        t_ranges.append((time.perf_counter() - t) / 5)
        off += g.copy_stage_to_device(engine.STAGE_WP_PRESMOOTH, buf.value + 24 * off, W - off)
        c = g.waypoint_counts()
        counts = c if counts is None else counts + c
        engines.append(g)
    fin = engines[0]
    fin.finish_path_async(buf.value, off, counts); fin.sync()
    t = time.perf_counter()
    for _ in range(5):
        fin.finish_path_async(buf.value, off, counts); fin.sync()
    t_fin = (time.perf_counter() - t) / 5
    # all ranges side by side on ONE GPU (each on its handle's own stream), then the finish: what splitting one large
    # cloud across streams of the same device would give
    t_conc = None
    if os.environ.get("PPP_SLICES_CONCURRENT", "0") == "1":
        def together():
            for g in engines:
                g.run_async()
            for g in engines:
                g.sync()
            o = 0
            for g in engines:
                o += g.copy_stage_to_device(engine.STAGE_WP_PRESMOOTH, buf.value + 24 * o, W - o)
            fin.finish_path_async(buf.value, o, counts); fin.sync()
        together(); together()
        t = time.perf_counter()
        for _ in range(5):
            together()
        t_conc = (time.perf_counter() - t) / 5
    same = off == W and fin.waypoints().tobytes() == one.waypoints().tobytes() and np.array_equal(fin.tail_index(), one.tail_index())
    if t_conc is not None:
        print("all %d ranges side by side on this one GPU + copies + finish (host-driven): %.3f ms" % (len(engines), t_conc * 1e3))
    print("%s: N %d S %d W %d | one handle %.3f ms | %d ranges: slowest %.3f ms, mean %.3f ms | finish on rank 0 %.3f ms | "
          "projected %d-GPU step (no gather) %.3f ms | identical to one handle: %s"
          % (name, len(pts), S, W, t_one * 1e3, len(t_ranges), max(t_ranges) * 1e3, np.mean(t_ranges) * 1e3, t_fin * 1e3,
             world, (max(t_ranges) + t_fin) * 1e3, same))
    print("per range ms:", " ".join("%.3f" % (x * 1e3) for x in t_ranges))
    assert same


if __name__ == "__main__":
    main()
