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
}


def run_case(plate, params):
    pts = synth.make_plate(**plate)
    o = ppo.Oracle(pts, **params)
    S = o.gen_path()
    assert S > 2, S
    W = o.get_path()
    out = dict(cloud=pts, S=S, W=W, px=o.slice_positions(), waypoints=o.waypoints(), tail=o.tail_index(),
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
    for name, (plate, params) in CASES.items():
        out = run_case(plate, params)
        out["params_keys"] = np.array(list(params.keys()))
        out["params_vals"] = np.array([float(v) for v in params.values()])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "N", len(out["cloud"]), "S", out["S"], "W", out["W"])
