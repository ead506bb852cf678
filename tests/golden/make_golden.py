"""Generates tests/golden/*.npz from the CPU oracle (oracle/ppp_oracle.cpp).

The reference ships no golden vectors and cannot be built in this image (PARITY UNPINNED), so
these fixtures pin the ORACLE: they are regression vectors for the restatement and the data the
GPU parity tests replay.  Each file holds the input cloud (metres, as in a PCD), the parameters,
and the expected outputs of every stage.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ppo  # noqa: E402
from polishpathplanning_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (plate args, params)
    "wavy_kd_center": (dict(nx=90, ny=48, kind="wavy", amp=8.0, seed=11), dict(tool_radius=6.0, pairing=0, walk=1)),
    "dome_brute_v1": (dict(nx=80, ny=44, kind="dome", amp=6.0, seed=12), dict(tool_radius=7.5, pairing=1, walk=3)),
    "blade_kd_sdir_trim5": (dict(nx=84, ny=52, kind="blade", amp=5.0, seed=13), dict(tool_radius=5.0, pairing=0, walk=2, trim=5.0)),
    "flat_kd_sectpath": (dict(nx=70, ny=40, kind="flat", amp=0.0, seed=14), dict(tool_radius=6.0, pairing=0, walk=0)),
    # the "next" rows: dynamic adjustment, and the constructors' preprocessing in front of the plan
    "wavy_kd_center_dynamic": (dict(nx=90, ny=48, kind="wavy", amp=8.0, seed=15), dict(tool_radius=6.0, pairing=0, walk=1, dynamic_adjustment=1)),
    "tilted_aligned_sor": (dict(nx=96, ny=40, kind="dome", amp=3.0, seed=16), dict(tool_radius=6.0, pairing=0, walk=1),
                           ["tilt", "align", "sor"]),
    "wavy_voxel_mls": (dict(nx=88, ny=46, kind="wavy", amp=6.0, seed=17), dict(tool_radius=6.0, pairing=0, walk=0), ["vox", "mls"]),
}

# a fixed tilt for the aligned case: leaves the plate's thin axis third after trans2center (a plan that makes sense)
TILT = (0.31, -0.22, 0.40)


def tilt(pts):
    ax, ay, az = TILT
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    return (pts.astype(np.float64) @ R.T + np.array([0.3, -0.2, 0.8])).astype(np.float32)


def apply_pre(o, ops):
    """the preprocessing calls of a case on an oracle / engine object (same method names on both)"""
    for op in ops:
        if op == "align":
            r = o.trans2center()
            assert (r[0] == 0) if isinstance(r[0], (int, np.integer)) else True
        elif op == "sor":
            o.remove_outlier(50, 1.0)
        elif op == "vox":
            o.voxel_down(0.1, 1.0, 1.0)
        elif op == "mls":
            o.smooth_mls(15.0, 3)


def run_case(plate, params, pre=()):
    pts = synth.make_plate(**plate)
    if "tilt" in pre:
        pts = tilt(pts)
    if "sor" in pre:   # a few points floating above the sheet for the filter to remove
        rng = np.random.default_rng(plate["seed"])
        fly = pts[rng.integers(0, len(pts), 25)].copy()
        fly[:, 2] += rng.uniform(0.006, 0.03, 25).astype(np.float32)
        pts = np.concatenate([pts, fly])
    o = ppo.Oracle(pts, **params)
    apply_pre(o, pre)
    cloud_pre = o.points()
    S = o.gen_path()
    assert S > 2, S
    W = o.get_path()
    out = dict(cloud=pts, cloud_pre=cloud_pre, pre_ops=np.array([p for p in pre if p != "tilt"], dtype="U8"), S=S, W=W, px=o.slice_positions(), waypoints=o.waypoints(), tail=o.tail_index(),
               wp_xyz=o.waypoints_xyz(), wp_nn=o.waypoint_nn(), wp_normals=o.waypoint_normals(),
               presmooth=o.waypoints_presmooth(), smoothed=o.waypoints_smoothed(), sweeps=o.smooth_sweeps())
    mn, mx = o.minmax()
    out["mn"], out["mx"] = mn, mx
    ny, nz, nc, idx, ic = [], [], [], [], []
    for s in range(S):
        y, x, z = o.nodes(s)
        ny.append(y); nz.append(z); nc.append(len(y))
        i = o.slice_indices(s)
        idx.append(i); ic.append(len(i))
    out["node_y"] = np.concatenate(ny); out["node_z"] = np.concatenate(nz); out["node_cnt"] = np.array(nc)
    out["slice_idx"] = np.concatenate(idx); out["slice_cnt"] = np.array(ic)
    return out


if __name__ == "__main__":
    for name, case in CASES.items():
        plate, params = case[0], case[1]
        out = run_case(plate, params, case[2] if len(case) > 2 else ())
        out["params_keys"] = np.array(list(params.keys()))
        out["params_vals"] = np.array([float(v) for v in params.values()])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "N", len(out["cloud"]), "S", out["S"], "W", out["W"])
