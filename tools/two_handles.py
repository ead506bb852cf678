#!/usr/bin/env python3
"""Steps of the headline workload enqueued on N handles taking turns (each handle = its own stream and buffers, the same resident
cloud on each): what overlapping consecutive, independent passes is worth against one handle's back-to-back steps.
usage: python tools/two_handles.py [config] [steps] [--dynamic]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polishpathplanning_amd import engine, synth
if os.environ.get("PPP_LIB"):   # a test build of the engine instead of the product library
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), os.environ["PPP_LIB"])
args = [a for a in sys.argv[1:] if a != "--dynamic"]
dyn = 1 if "--dynamic" in sys.argv else 0          # Dynamic_adjustment = true: a chain of dependent steps that fills a quarter of the chip
name = args[0] if len(args) > 0 else "cfg2_1m_s256"
steps = int(args[1]) if len(args) > 1 else 200
pts, cfg = synth.make_config(name)
for nh in ((1, 2, 3, 4, 6, 8) if dyn else (1, 2, 3, 4)):
    hs = [engine.Engine(0, tool_radius=cfg["tool_radius"], dynamic_adjustment=dyn) for _ in range(nh)]
    for h in hs:
        if not os.environ.get("PPP_NO_SIDE_BY_SIDE"):
            h.set_side_by_side(nh)      # (tell the plan how many passes share the device)
        h.set_cloud(pts)
        for _ in range(3):
            h.run_async()
        h.sync()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for k in range(steps):
            hs[k % nh].run_async()
        for h in hs:
            h.sync()
        best = min(best, (time.perf_counter() - t) / steps * 1e3)
    W = hs[0].num_waypoints()
    print("%s%s: %d handle(s) taking turns: %.4f ms per step, %.3e waypoints/s" % (name, " with Dynamic_adjustment" if dyn else "", nh, best, W / (best * 1e-3)), flush=True)
    for h in hs:
        h.close()
