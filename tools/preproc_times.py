#!/usr/bin/env python3
"""Wall-clock of the cloud preprocessing entry points on the GPU for one synthetic cloud:
   remove_outlier (SOR 50 / 1 sigma), trans2center, voxel_down (main.cpp:25's 0.1 x 1 x 1 and a 3 mm cube), MLS smooth (order 3, r 15).
   usage: preproc_times.py [config name, default cfg2_1m_s256]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polishpathplanning_amd import engine, synth

def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "cfg2_1m_s256"
    pts, cfg = synth.make_config(name)
    print("cloud %s: %d points" % (name, len(pts)))
    def fresh():
        e = engine.Engine(0, tool_radius=6.0)
        e.set_cloud(pts)
        e.nearest(pts[:1] * 1000)   # builds the slab index outside the timings
        return e
    for label, fn in [("remove_outlier(50, 1.0)", lambda e: e.remove_outlier(50, 1.0)),
                      ("voxel_down(0.1, 1, 1)", lambda e: e.voxel_down(0.1, 1.0, 1.0)),
                      ("voxel_down(3, 3, 3)", lambda e: e.voxel_down(3.0, 3.0, 3.0)),
                      ("smooth_mls(15, 3)", lambda e: e.smooth_mls(15.0, 3)),
                      ("smooth_mls(15, 2)", lambda e: e.smooth_mls(15.0, 2)),
                      ("trans2center()", lambda e: e.trans2center()[1])]:
        best = 1e9
        res = None
        for rep in range(3):
            e = fresh()
            t0 = time.perf_counter()
            res = fn(e)
            best = min(best, time.perf_counter() - t0)
        print("%-26s %9.3f ms   -> %s" % (label, best * 1e3, res))


if __name__ == "__main__":
    main()
