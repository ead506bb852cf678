#!/usr/bin/env python3
"""File to file: what a planner per workpiece costs end to end (PCD on disk -> pathFile on disk) when several workpieces go
through one process (examples/workpieces.cpp), with the planners sharing engine handles (ppp::HandlePool) and without.
   python tools/workpieces_flow.py [--config cfg2_1m_s256] [--count 5] [--format binary|ascii|compressed]
Writes its clouds into a temporary directory; prints the example's report lines for both runs."""
import argparse, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polishpathplanning_amd import engine, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2_1m_s256")
    ap.add_argument("--count", type=int, default=5)
    ap.add_argument("--format", default="binary", choices=["binary", "ascii", "compressed"])
    ap.add_argument("--dynamic", action="store_true", help="Dynamic_adjustment = true (the reference's config.txt default)")
    a = ap.parse_args()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "workpieces"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config(a.config)
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as d:
        names = []
        for i in range(a.count):      # the same plate scanned again: other point order, a little lower every time
            c = pts[rng.permutation(len(pts))] - np.float32([0, 0, 1e-5 * i])
            names.append(os.path.join(d, "w%d.pcd" % i))
            engine.save_pcd(names[-1], np.ascontiguousarray(c, np.float32), binary={"binary": True, "ascii": False, "compressed": "compressed"}[a.format])
        out = os.path.join(d, "WayPoints.txt")
        conf = os.path.join(d, "config.txt")
        open(conf, "w").write("Tool_Radius = %g\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\n"
                              "Smooth = false\nAlignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = %s\n"
                              "Adjust_Threshold = 1\ntoolthickness = 10\ndepth = 0.01\n" % (cfg.get("tool_radius", 6.0), out, "true" if a.dynamic else "false"))
        for label, extra in (("handle pool", {}), ("no pool", {"PPP_NO_HANDLE_POOL": "1"})):
            r = subprocess.run([os.path.join(ROOT, "examples", "workpieces")] + names, env=dict(os.environ, PPP_CONFIG=conf, **extra),
                               capture_output=True, text=True, timeout=600, cwd=d)
            if r.returncode != 0:
                print(r.stdout[-2000:], r.stderr[-2000:])
                raise SystemExit("workpieces failed")
            print("%s, %s, %d x %s%s:" % (label, a.format, a.count, a.config, ", Dynamic_adjustment = true" if a.dynamic else ""))
            for ln in r.stdout.splitlines():
                if ln.startswith("workpieces:"):
                    print("   " + ln)


if __name__ == "__main__":
    main()
