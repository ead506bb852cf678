"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle).

CPU: the oracle still reproduces every stage bit for bit.  GPU: the HIP engine, driven through
the C ABI, reproduces the same vectors (bit-exact for index / node / count stages, <= 1e-4 m
for the floating-point waypoint list)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
INT_KEYS = ("pairing", "walk", "dynamic_adjustment")


def load(path):
    g = np.load(path)
    params = {k: (int(v) if k in INT_KEYS else float(v)) for k, v in zip(g["params_keys"], g["params_vals"])}
    return g, params


def preprocess(obj, g):
    """the constructors' preprocessing recorded with the fixture (smooth / align / remove; v1's voxel grid), then the cloud check"""
    ops = [str(x) for x in g["pre_ops"]] if "pre_ops" in g else []
    for op in ops:
        if op == "align":
            obj.trans2center()
        elif op == "sor":
            obj.remove_outlier(50, 1.0)
        elif op == "vox":
            obj.voxel_down(0.1, 1.0, 1.0)
        elif op == "mls":
            obj.smooth_mls(15.0, 3)
    return ops


def cloud_matches(got, want, ops):
    """bit for bit; after MLS (f64 sums grouped differently on the device, libm's exp) to one float ulp.  Returns exactness."""
    got, want = np.nan_to_num(got), np.nan_to_num(want)
    assert got.shape == want.shape
    if np.array_equal(got, want):
        return True
    assert "mls" in ops
    assert (np.abs(got - want) <= np.spacing(np.maximum(np.abs(got), np.abs(want)).astype(np.float32))).all()
    return False


def test_fixtures_exist():
    assert len(GOLD) >= 7


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(oracle_mod, path):
    g, params = load(path)
    o = oracle_mod.Oracle(g["cloud"], **params)
    ops = preprocess(o, g)
    if ops:
        assert cloud_matches(o.points(), g["cloud_pre"], ops)
    assert o.gen_path() == int(g["S"]) and o.get_path() == int(g["W"])
    assert np.array_equal(o.slice_positions(), g["px"])
    assert np.array_equal(o.waypoints(), g["waypoints"])
    assert np.array_equal(o.tail_index(), g["tail"])
    assert np.array_equal(o.waypoint_nn(), g["wp_nn"])
    off = np.concatenate([[0], np.cumsum(g["node_cnt"])])
    ioff = np.concatenate([[0], np.cumsum(g["slice_cnt"])])
    for s in range(int(g["S"])):
        y, x, z = o.nodes(s)
        assert np.array_equal(y, g["node_y"][off[s]:off[s + 1]]) and np.array_equal(z, g["node_z"][off[s]:off[s + 1]])
        assert np.array_equal(o.slice_indices(s), g["slice_idx"][ioff[s]:ioff[s + 1]])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_engine_reproduces_golden(engine_mod, path):
    g, params = load(path)
    e = engine_mod.Engine(0, **params)
    e.set_cloud(g["cloud"])
    ops = preprocess(e, g)
    exact = cloud_matches(e.cloud(), g["cloud_pre"], ops) if ops else True
    assert e.gen_path() == int(g["S"])
    assert e.get_path() == int(g["W"])
    if not exact:   # a smoothed cloud that differs in a last bit: the list within the tolerance, the stages not bit for bit
        wp = e.waypoints()
        assert np.linalg.norm(wp[:, :3] - g["waypoints"][:, :3], axis=1).max() <= 1e-4
        return
    mn, mx = e.minmax()
    assert np.array_equal(mn, g["mn"]) and np.array_equal(mx, g["mx"])
    assert np.array_equal(e.slice_positions(), g["px"])
    off = np.concatenate([[0], np.cumsum(g["node_cnt"])])
    ioff = np.concatenate([[0], np.cumsum(g["slice_cnt"])])
    for s in range(int(g["S"])):
        y, x, z = e.nodes(s)
        assert np.array_equal(y, g["node_y"][off[s]:off[s + 1]]) and np.array_equal(z, g["node_z"][off[s]:off[s + 1]])
        if not params.get("dynamic_adjustment"):
            assert np.all(x == np.float64(g["px"][s]))
        assert np.array_equal(e.slice_indices(s), g["slice_idx"][ioff[s]:ioff[s + 1]])
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_XYZ), g["wp_xyz"])
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_NN), g["wp_nn"])
    assert np.array_equal(e.tail_index(), g["tail"])
    wp = e.waypoints()
    assert np.linalg.norm(wp[:, :3] - g["waypoints"][:, :3], axis=1).max() <= 1e-4  # north_star tolerance, metres
    d = np.abs(wp[:, 3:] - g["waypoints"][:, 3:])
    assert np.minimum(d, np.abs(d - 2 * np.pi)).max() <= 1e-4
