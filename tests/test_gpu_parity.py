"""Parity of the HIP hot path (through the C ABI) with the CPU oracle on the same seeded inputs.

Bar: bit-exact for every integer / index / count stage and for the float stages that only
involve the reference's own float operations (min/max, plane positions, band index lists,
spline knots, sampled waypoints, nearest-neighbour ids); <= 1e-4 m (north_star) for the final
floating-point waypoint list, whose normals / trigonometry go through different libm's.
"""
import numpy as np
import pytest

from polishpathplanning_amd import synth

pytestmark = pytest.mark.gpu

TOL_M = 1e-4      # BASELINE.json north_star: "within 1e-4 m"
TOL_RAD = 1e-4


def run_pair(engine_mod, oracle_mod, pts, **params):
    o = oracle_mod.Oracle(pts, **params)
    e = engine_mod.Engine(0, **params)
    e.set_cloud(pts)
    return e, o


def assert_full_parity(engine_mod, e, o, every_slice=True, unit=1.0):
    """unit: size of a metre in the waypoint list's length unit (1000 when ChangeRange is off)."""
    So = o.gen_path(); S = e.gen_path()
    assert S == So
    Wo = o.get_path(); W = e.get_path()
    assert W == Wo
    mn, mx = e.minmax(); omn, omx = o.minmax()
    assert np.array_equal(mn, omn) and np.array_equal(mx, omx)
    assert np.array_equal(e.slice_positions(), o.slice_positions())
    step = 1 if every_slice else max(1, S // 24)
    for s in range(0, S, step):
        assert np.array_equal(e.slice_indices(s), o.slice_indices(s)), s
    for s in range(S):
        gy, gx, gz = e.nodes(s); oy, ox, oz = o.nodes(s)
        assert np.array_equal(gy, oy) and np.array_equal(gx, ox) and np.array_equal(gz, oz), s
    assert np.array_equal(e.tail_index(), o.tail_index())
    if W == 0:
        return
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_XYZ), o.waypoints_xyz())
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_NN), o.waypoint_nn())
    n, on = e.stage(engine_mod.STAGE_WP_NORMAL), o.waypoint_normals()
    ang = np.arctan2(np.linalg.norm(np.cross(n[:, :3], on[:, :3]), axis=1), np.sum(n[:, :3] * on[:, :3], axis=1))
    assert ang.max() < 1e-4
    pre, opre = e.stage(engine_mod.STAGE_WP_PRESMOOTH), o.waypoints_presmooth()
    assert np.abs(pre[:, :3] - opre[:, :3]).max() <= 1e-6 * unit
    sm, osm = e.stage(engine_mod.STAGE_WP_SMOOTHED), o.waypoints_smoothed()
    # the GPU solves the sweeps' fixed point directly (ppp_kernels.h a13); the oracle sweeps sequentially in float until the
    # list is stationary: they differ by the float storage floor (1.2e-7 m typical, 6e-7 m on 3..5-waypoint lists)
    assert np.abs(sm[:, :3] - osm[:, :3]).max() <= 1e-6 * unit
    wp, owp = e.waypoints(), o.waypoints()
    assert np.linalg.norm(wp[:, :3] - owp[:, :3], axis=1).max() <= TOL_M * unit
    d = np.abs(wp[:, 3:] - owp[:, 3:])
    assert np.minimum(d, np.abs(d - 2 * np.pi)).max() <= TOL_RAD


@pytest.mark.parametrize("name,pairing,walk", [
    ("tiny_5k", 0, 1), ("tiny_5k", 1, 1), ("small_40k", 0, 1), ("small_40k", 1, 3), ("small_40k", 0, 0),
    ("small_40k", 0, 2), ("small_40k", 1, 4), ("cfg1_50k_s32", 0, 1), ("cfg1_50k_s32", 1, 3),
])
def test_pipeline_parity(engine_mod, oracle_mod, name, pairing, walk):
    pts, cfg = synth.make_config(name)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=cfg["tool_radius"], pairing=pairing, walk=walk)
    assert_full_parity(engine_mod, e, o)


@pytest.mark.parametrize("name", ["cfg3_250k_s128", "cfg2_1m_s256"])
def test_baseline_configs_full_size(engine_mod, oracle_mod, name):
    """BASELINE.json sizes: the oracle's fast mode still finishes in about a second."""
    pts, cfg = synth.make_config(name)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=cfg["tool_radius"])
    assert_full_parity(engine_mod, e, o, every_slice=False)
    assert e.num_slices() == cfg["slices"]


def test_contour_variant_trim5_no_drop_no_smooth(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0, trim=5.0, drop_ends=0, smooth=0, rpy_resolution=2.0)
    assert_full_parity(engine_mod, e, o)
    assert e.smooth_sweeps() == 0 == o.smooth_sweeps()


def test_other_resolutions(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=9.0, path_resolution=3.3, rpy_resolution=4.0, ee_length=0.12)
    assert_full_parity(engine_mod, e, o)


def test_not_change_range(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("tiny_5k")
    mm = (pts * 1000).astype(np.float32)
    e, o = run_pair(engine_mod, oracle_mod, mm, tool_radius=6.0, change_range=0)
    assert_full_parity(engine_mod, e, o, unit=1000.0)  # the list is in millimetres here


def test_rerun_is_bitwise_reproducible(engine_mod):
    pts, cfg = synth.make_config("small_40k")
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    e.gen_path(); e.get_path()
    a = e.waypoints().tobytes()
    for _ in range(3):
        e.gen_path(); e.get_path()
        assert e.waypoints().tobytes() == a
    e2 = engine_mod.Engine(0, tool_radius=6.0)
    e2.set_cloud(pts[:, :3].copy())
    e2.gen_path(); e2.get_path()
    assert e2.waypoints().tobytes() == a


def test_point_order_changes_nothing_but_indices(engine_mod):
    """Property: a permutation of the cloud permutes indices only (no exact ties in the fixtures)."""
    pts, cfg = synth.make_config("tiny_5k")
    perm = np.random.default_rng(0).permutation(len(pts))
    e1 = engine_mod.Engine(0, tool_radius=6.0); e1.set_cloud(pts); e1.gen_path(); e1.get_path()
    e2 = engine_mod.Engine(0, tool_radius=6.0); e2.set_cloud(pts[perm]); e2.gen_path(); e2.get_path()
    assert np.array_equal(e1.stage(engine_mod.STAGE_WP_XYZ), e2.stage(engine_mod.STAGE_WP_XYZ))
    assert np.array_equal(perm[e2.stage(engine_mod.STAGE_WP_NN)], e1.stage(engine_mod.STAGE_WP_NN))
    assert np.abs(e1.waypoints() - e2.waypoints()).max() < 1e-5


@pytest.mark.parametrize("axis", [0, 1])
def test_scan_ordered_clouds(engine_mod, oracle_mod, axis):
    """The synthetic clouds are randomly permuted (the reference's results depend on index order); a scanner delivers rows.  The same
    plate sorted by x (rows along y: a binning workgroup's points fall into one or two windows) and by y: full parity with the oracle
    on the window path, nothing handed back."""
    pts, cfg = synth.make_config("small_40k")
    pts = np.ascontiguousarray(pts[np.argsort(pts[:, axis], kind="stable")])
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=cfg["tool_radius"])
    assert_full_parity(engine_mod, e, o)
    assert e.fast_path()


def test_stride_32_pointxyzrgb_layout(engine_mod):
    pts, cfg = synth.make_config("tiny_5k")
    aos = np.zeros((len(pts), 8), np.float32)  # pcl::PointXYZRGB: xyz + pad, rgba + pad
    aos[:, :3] = pts
    aos[:, 3] = 1.0
    aos[:, 4] = np.float32(2.3e-38)
    a = engine_mod.Engine(0, tool_radius=6.0); a.set_cloud(pts); a.gen_path(); a.get_path()
    b = engine_mod.Engine(0, tool_radius=6.0); b.set_cloud(aos); b.gen_path(); b.get_path()
    assert a.waypoints().tobytes() == b.waypoints().tobytes()


# ---------------- API mirrors of the reference's public methods ----------------
def test_ranged_x_index_api(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    mn, mx = o.minmax()
    for pos in [int(mn[0]) - 5, int(mn[0]), 77, 78, 300, int(mx[0]) + 1, int(mx[0]) + 9]:
        assert np.array_equal(e.ranged_x_index(pos), o.ranged_x_index(pos)), pos


@pytest.mark.parametrize("pairing", [0, 1])
def test_insert_point_api_arbitrary_index_order(engine_mod, oracle_mod, pairing):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0, pairing=pairing)
    rng = np.random.default_rng(3)
    for px in [100.0, 250.6, 411.2]:
        idx = o.ranged_x_index(int(px))
        for order in (idx, idx[::-1].copy(), rng.permutation(idx)):  # El/Er order is the caller's order
            m, y, x, z = e.insert_point(order, px)
            om, oy, ox, oz = o.insert_point(order, px)
            assert m == om and np.array_equal(y, oy) and np.array_equal(x, ox) and np.array_equal(z, oz)


def test_insert_point_api_errors(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("tiny_5k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    idx = o.ranged_x_index(40)
    assert e.insert_point(idx, -500.0)[0] == engine_mod.ERR_SLICE  # empty right side
    m, y, x, z = e.insert_point(idx, 5000.0)                        # empty left side: empty map
    assert m == 0
    assert e.insert_point(np.zeros(0, np.int32), 40.0)[0] == 0


def test_eval_spline_api(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    e.gen_path(); o.gen_path()
    for s in [0, 7, e.num_slices() - 1]:
        y, x, z = o.nodes(s)
        q = np.concatenate([y[:5], np.linspace(y[0], y[-1], 257), [y[-1]]])
        rc, got = e.eval_spline(s, q)
        orc, want = o.eval_spline(s, q)
        assert rc == 0 == orc and np.array_equal(got, want)   # same double operations, same order
        rc, got = e.eval_spline(s, np.array([y[0] - 1e-9, y[-1] + 1.0, y[3]]))
        assert rc == engine_mod.ERR_DOMAIN and np.isnan(got[0]).all() and np.isnan(got[1]).all() and not np.isnan(got[2]).any()


def test_spline_class_on_caller_knots(engine_mod, oracle_mod):
    """include/Spline.h:10-42: Spline(number, y, x, z), point(), miny/bigy, restart() on knots the CALLER supplies (what
    OnePath / path_track / dynamic_adjust_path do), against the oracle's steffen.c restatement: same doubles."""
    rng = np.random.default_rng(21)
    for n in (3, 4, 17, 600):
        y = np.cumsum(rng.uniform(0.05, 3.0, n)) - 40.0
        x = 100.0 + np.cumsum(rng.normal(0, 0.3, n)); z = 1500.0 + 30 * np.sin(y / 25.0) + rng.normal(0, 0.2, n)
        sp = engine_mod.Spline(y, x, z)
        assert sp.range() == (y[0], y[-1], n)
        q = np.concatenate([y, np.linspace(y[0], y[-1], 1001), (y[:-1] + y[1:]) / 2])
        rc, got = sp.point(q)
        assert rc == 0
        assert np.array_equal(got[:, 0], oracle_mod.steffen(y, x, q)[1]) and np.array_equal(got[:, 2], oracle_mod.steffen(y, z, q)[1])
        assert np.array_equal(got[:, 1], q)
        assert np.array_equal(got[:n, 0], x) and np.array_equal(got[:n, 2], z)          # interpolates its knots
        rc, got = sp.point([y[0] - 1e-9, y[-1] + 1.0, np.nan, y[1]])                     # GSL_EDOM
        assert rc == engine_mod.ERR_DOMAIN and np.isnan(got[:3]).all() and not np.isnan(got[3]).any()
        # restart(): dynamic_adjust_path re-fits the same object on the adjusted knots (path_dynamic_alg.cpp:297-303)
        y2 = np.cumsum(rng.uniform(0.1, 2.0, n + 5)); x2 = rng.normal(0, 1, n + 5); z2 = rng.normal(0, 1, n + 5)
        sp.restart(y2, x2, z2)
        assert sp.range() == (y2[0], y2[-1], n + 5)
        q2 = np.linspace(y2[0], y2[-1], 333)
        rc, got = sp.point(q2)
        assert rc == 0 and np.array_equal(got[:, 0], oracle_mod.steffen(y2, x2, q2)[1]) and np.array_equal(got[:, 2], oracle_mod.steffen(y2, z2, q2)[1])
        sp.close()
    # where GSL raises GSL_EINVAL and aborts: fewer than 3 knots, y not strictly increasing
    for bad_y in ([0.0, 1.0], [0.0, 1.0, 1.0, 2.0], [0.0, 2.0, 1.0]):
        with pytest.raises(engine_mod.PPPError) as ei:
            engine_mod.Spline(bad_y, bad_y, bad_y)
        assert ei.value.code == engine_mod.ERR_ARG
    # the same knots through a planner slice and through the class give the same curve
    pts, cfg = synth.make_config("tiny_5k")
    e = engine_mod.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.gen_path()
    ky, kx, kz = e.nodes(3)
    sp = engine_mod.Spline(ky, kx, kz)
    q = np.linspace(ky[0], ky[-1], 77)
    assert np.array_equal(sp.point(q)[1], e.eval_spline(3, q)[1])


def test_nearest_and_normals_api(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    cloud = o.points()
    rng = np.random.default_rng(5)
    q = cloud[rng.integers(0, len(cloud), 500)] + rng.normal(0, 1.5, (500, 3)).astype(np.float32)
    q[:5] += 40.0  # far queries: the slab walk must keep going
    got = e.nearest(q)
    want = np.array([o.nearest(v)[0] for v in q])
    assert np.array_equal(got, want)
    idx = rng.integers(0, len(cloud), 400).astype(np.int32)
    n = e.normals_at(idx)
    on = np.stack([o.normal_at(i) for i in idx])
    nan = np.isnan(on[:, 0])
    assert np.array_equal(np.isnan(n[:, 0]), nan)
    ang = np.arctan2(np.linalg.norm(np.cross(n[~nan, :3], on[~nan, :3]), axis=1), np.sum(n[~nan, :3] * on[~nan, :3], axis=1))
    assert ang.max() < 1e-4 and np.abs(n[~nan, 3] - on[~nan, 3]).max() < 1e-5


# ---------------- edge cases the domain has ----------------
def test_nan_points_are_dropped_everywhere(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("tiny_5k")
    pts = pts.copy()
    pts[::97, 0] = np.nan
    pts[5::131, 2] = np.inf
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    assert_full_parity(engine_mod, e, o)


def test_slice_with_too_few_nodes_is_reported_not_aborted(engine_mod, oracle_mod):
    # a cloud whose first band holds a single column of points: the reference would abort in GSL
    pts = synth.make_plate(40, 30, seed=2)
    keep = ~((pts[:, 0] * 1000 < 9.0) & (np.abs(pts[:, 1]) * 1000 > 3.0))
    pts = pts[keep]
    o = oracle_mod.Oracle(pts, tool_radius=3.0, walk=3)
    rc = o.gen_path()
    e = engine_mod.Engine(0, tool_radius=3.0, walk=3)
    e.set_cloud(pts)
    e.gen_path_async()
    if rc < 0:
        with pytest.raises(engine_mod.PPPError) as ei:
            e.sync()
        assert ei.value.code == engine_mod.ERR_SLICE and e.failed_slice() == -rc - 1
    else:
        e.sync()


def test_empty_cloud(engine_mod):
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(np.zeros((0, 3), np.float32))
    assert e.gen_path() == 0
    assert e.get_path() == 0
    assert e.waypoints().shape == (0, 6)


def test_short_slices_take_the_sequential_rpy_path(engine_mod, oracle_mod):
    # a narrow plate: every slice yields fewer than RPYres+1 waypoints (App. B.6 overlap quirk)
    pts = synth.make_plate(120, 38, kind="wavy", amp=3.0, seed=7)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    assert_full_parity(engine_mod, e, o)
    assert o.get_path() > 0 and np.diff(np.concatenate([[-1], o.tail_index()])).max() <= 7


def test_unsupported_options_fail_loudly(engine_mod):
    e = engine_mod.Engine(0)
    with pytest.raises(engine_mod.PPPError) as ei:
        e.set_params(dynamic_adjustment=1, walk=0)   # SectPath::GenPath has no adjustment (path_slicing_alg.cpp:290-342)
    assert ei.value.code == engine_mod.ERR_UNSUPPORTED
    with pytest.raises(engine_mod.PPPError):
        e.set_params(alignment=1, dynamic_adjustment=0, walk=1)
    with pytest.raises(engine_mod.PPPError):
        e.set_params(alignment=0, path_resolution=0.0)


def test_call_order_errors(engine_mod):
    e = engine_mod.Engine(0)
    with pytest.raises(engine_mod.PPPError):
        e.gen_path_async()          # no cloud
    pts, cfg = synth.make_config("tiny_5k")
    e.set_cloud(pts)
    with pytest.raises(engine_mod.PPPError):
        e.get_path_async()          # getPath before GenPath


def test_params_change_replans(engine_mod, oracle_mod):
    pts, cfg = synth.make_config("small_40k")
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    e.gen_path(); e.get_path()
    e.set_params(tool_radius=10.0, walk=0)
    o = oracle_mod.Oracle(pts, tool_radius=10.0, walk=0)
    assert_full_parity(engine_mod, e, o)


def test_two_handles_interleaved(engine_mod, oracle_mod):
    a_pts, _ = synth.make_config("tiny_5k")
    b_pts, _ = synth.make_config("small_40k")
    a = engine_mod.Engine(0, tool_radius=6.0); b = engine_mod.Engine(0, tool_radius=6.0)
    a.set_cloud(a_pts); b.set_cloud(b_pts)
    a.gen_path_async(); b.gen_path_async(); a.get_path_async(); b.get_path_async()
    a.sync(); b.sync()
    for eng, pts in ((a, a_pts), (b, b_pts)):
        o = oracle_mod.Oracle(pts, tool_radius=6.0); o.gen_path(); o.get_path()
        assert np.linalg.norm(eng.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1).max() <= TOL_M


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize("dynamic,remove,smooth,align", [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 1, 1, 0), (0, 1, 0, 1), (1, 0, 0, 1)])
def test_connect_cli_end_to_end(engine_mod, oracle_mod, tmp_path, dynamic, remove, smooth, align):
    """The drop-in C++ classes (include/Path_Generate_Algorithm.h) driven like src/connect.cpp:
    PCD in, pathFile out; with and without Dynamic_adjustment (config.txt:13), RemoveOutlier (config.txt:11) and
    Smooth (config.txt:9, with its smooth_<name> side file) and Alignment (config.txt:10, on a tilted plate)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "connect", "connect1", "main"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("small_40k")
    if align:
        pts = _rotated_plate(7)
    if remove:   # a few points floating above the sheet
        rng = np.random.default_rng(9)
        fly = pts[rng.integers(0, len(pts), 30)].copy()
        fly[:, 2] += rng.uniform(0.005, 0.03, 30).astype(np.float32)
        pts = np.concatenate([pts, fly])
    pcd = str(tmp_path / "workpiece.pcd")
    engine_mod.save_pcd(pcd, pts, binary="compressed" if remove else True)
    out = str(tmp_path / "WayPoints.txt")
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 6\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\n"
                    "Smooth = %s\nAlignment = %s\nChangeRange = true\nRemoveOutlier = %s\nDynamic_adjustment = %s\n"
                    "Adjust_Threshold = 1\ntoolthickness = 10\ndepth = 0.01\n"
                    % (out, "true" if smooth else "false", "true" if align else "false", "true" if remove else "false", "true" if dynamic else "false"))
    env = dict(os.environ, PPP_CONFIG=str(conf))
    if smooth:
        pcd = "workpiece.pcd"               # "smooth_" + name must be a writable path: run in the directory
    for exe, walk in (("connect", 1), ("connect1", 2)):
        if os.path.exists(out):
            os.remove(out)
        r = subprocess.run([os.path.join(root, "examples", exe), pcd], env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        assert "!!!!! GOT PATH !!!!!" in r.stdout
        got = np.loadtxt(out, dtype=np.float64).reshape(-1, 6)
        o = oracle_mod.Oracle(pts, tool_radius=6.0, walk=walk, dynamic_adjustment=dynamic)
        if smooth:                          # ctor order: smooth, (align,) remove (path_slicing_alg.cpp:27-29)
            assert len(pts) - 30 <= o.smooth_mls(15.0, 3) <= len(pts)   # flyers more than 15 mm from the sheet have no neighbours
            side = engine_mod.load_pcd(str(tmp_path / "smooth_workpiece.pcd"))[0]
            assert np.abs(side * 1000.0 - o.points()).max() < 2e-3      # ascii, metres
        if align:
            assert o.trans2center()[0] == 0
        if remove:
            assert o.remove_outlier(50, 1.0)[0] < len(pts)
        o.gen_path(); o.get_path()
        want = o.waypoints()
        assert got.shape == want.shape
        assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-4 + 5e-6 * np.abs(want[:, :3]).max()  # 6 significant digits in the file
    if not align:       # ./main never aligns (main.cpp:26 is commented out): it needs the plate in its own frame
        r = subprocess.run([os.path.join(root, "examples", "main"), pcd], env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0 and "number of paths" in r.stdout
        if smooth:      # the calls main.cpp:25-29 keeps commented out, switched on: voxel grid, slicing_method, MLS (+ its side file)
            env2 = dict(env, PPP_MAIN_VOXEL="1", PPP_MAIN_SLICING="1", PPP_MAIN_SMOOTH="1")
            r = subprocess.run([os.path.join(root, "examples", "main"), "workpiece.pcd"], env=env2, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
            assert r.returncode == 0 and r.stdout.count("number of paths") == 2, r.stdout + r.stderr
            assert os.path.exists(str(tmp_path / "smooth_workpiece.pcd"))


def test_set_cloud_pcd_streams_the_file_into_hbm(engine_mod, tmp_path):
    """ppp_set_cloud_pcd: `DATA binary` records with consecutive float32 x y z go to HBM in pieces through the handle's two pinned
    buffers (here 20 MB of 16-byte and 24-byte records: three pieces, x at offset 0 and at offset 4) and the conversion kernel
    picks the coordinates out of them; every other flavour takes the host loader.  The resident cloud must be, bit for bit,
    what ppp_set_cloud makes of ppp_load_pcd's points; the same handle then takes a small file and a refused one."""
    import os
    rng = np.random.default_rng(31)
    n = 1_300_000
    pts = (rng.random((n, 3), dtype=np.float32) * np.float32([3.0, 0.24, 0.05])).astype(np.float32)
    pts[5] = [np.nan, 0.1, 0.2]
    e = engine_mod.Engine()
    ref = engine_mod.Engine()

    def check(path, want_pts, vp_want=None):
        got_n, vp = e.set_cloud_pcd(path)
        xyz, vp_file = engine_mod.load_pcd(path)
        assert np.array_equal(xyz.view(np.uint32), np.ascontiguousarray(want_pts, np.float32).view(np.uint32))
        ref.set_cloud(xyz, viewpoint=vp_file[:3])
        assert got_n == len(want_pts) == e.n == ref.n
        a, b = e.cloud(), ref.cloud()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(vp, vp_file) and (vp_want is None or list(vp) == vp_want)
        assert all(np.array_equal(u, v) for u, v in zip(e.minmax(), ref.minmax()))

    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
    rec["x"], rec["y"], rec["z"], rec["rgb"] = pts[:, 0], pts[:, 1], pts[:, 2], 0x00ff00
    p = str(tmp_path / "xyzrgb.pcd")
    open(p, "wb").write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
                         "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0.5 -1 2 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n)).encode() + rec.tobytes())
    assert engine_mod.pcd_probe(p).record_bytes == 16
    check(p, pts, [0.5, -1, 2, 1, 0, 0, 0])
    rec2 = np.zeros(n, dtype=[("rgb", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("normal", "<f4", (2,))])
    rec2["x"], rec2["y"], rec2["z"] = pts[::-1, 0], pts[::-1, 1], pts[::-1, 2]
    p2 = str(tmp_path / "rgbxyz.pcd")
    open(p2, "wb").write(("VERSION 0.7\nFIELDS rgb x y z normal\nSIZE 4 4 4 4 4\nTYPE U F F F F\nCOUNT 1 1 1 1 2\nWIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA binary\n" % (n, n)).encode()
                         + rec2.tobytes())
    lay = engine_mod.pcd_probe(p2)
    assert (lay.record_bytes, lay.x_offset) == (24, 4)
    check(p2, pts[::-1])
    small = pts[:40_000]
    for mode in (True, False, "compressed"):          # packed binary (direct), ascii and compressed (host loader)
        q = str(tmp_path / ("small_%s.pcd" % mode))
        engine_mod.save_pcd(q, small, binary=mode)
        check(q, small)
    # a planner on the streamed cloud gives the list of a planner on the loaded one
    plate, cfg = synth.make_config("small_40k")
    q = str(tmp_path / "plate.pcd")
    engine_mod.save_pcd(q, plate, binary=True)
    for eng in (e, ref):
        eng.set_params(tool_radius=6.0)
    e.set_cloud_pcd(q); ref.set_cloud(plate)
    for eng in (e, ref):
        eng.gen_path(); eng.get_path()
    assert np.array_equal(e.waypoints().view(np.uint32), ref.waypoints().view(np.uint32))
    # refused files leave an error, not a half-loaded cloud
    open(p, "r+b").truncate(os.path.getsize(p) - 100)
    with pytest.raises(engine_mod.PPPError):
        e.set_cloud_pcd(p)
    with pytest.raises(engine_mod.PPPError):
        e.set_cloud_pcd(str(tmp_path / "missing.pcd"))
    e.set_cloud_pcd(q)
    e.gen_path(); e.get_path()
    assert np.array_equal(e.waypoints().view(np.uint32), ref.waypoints().view(np.uint32))


def test_planners_of_one_process_share_engine_handles(engine_mod, tmp_path):
    """A planner per workpiece, one after the other in one process (examples/workpieces.cpp): the second and third take the
    handle the one before gave back (ppp::HandlePool in ppp_planner.hpp).  Nothing of an earlier workpiece may show: every
    pathFile equals, byte for byte, the one a process of its own writes for that cloud -- a cloud of the same size as the last
    (plan inherited), a smaller one, and the first again."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "connect", "workpieces"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("small_40k")
    rng = np.random.default_rng(21)
    clouds = [pts, pts[rng.permutation(len(pts))] + np.float32([0, 0, 0.002]), pts[pts[:, 1] < np.median(pts[:, 1])], pts]
    names = []
    for i, c in enumerate(clouds):
        names.append(str(tmp_path / ("w%d.pcd" % i)))
        engine_mod.save_pcd(names[-1], np.ascontiguousarray(c, dtype=np.float32), binary=True)
    out = str(tmp_path / "WayPoints.txt")
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 6\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\n"
                    "Smooth = false\nAlignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = false\n"
                    "Adjust_Threshold = 1\ntoolthickness = 10\ndepth = 0.01\n" % out)
    env = dict(os.environ, PPP_CONFIG=str(conf))
    alone = []
    for nm in names:
        r = subprocess.run([os.path.join(root, "examples", "connect"), nm], env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        alone.append(open(out, "rb").read())
        os.remove(out)
    assert len(set(alone[:3])) == 3 and alone[0] == alone[3]
    r = subprocess.run([os.path.join(root, "examples", "workpieces")] + names, env=env, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 planned, 3 planners served from the handle pool" in r.stdout, r.stdout
    for i in range(4):
        assert open(out + ".%d" % i, "rb").read() == alone[i], i
    # and with the pool switched off: four handles made and destroyed, the same files
    r = subprocess.run([os.path.join(root, "examples", "workpieces")] + names, env=dict(env, PPP_NO_HANDLE_POOL="1"), capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0 and "4 planned, 0 planners served" in r.stdout, r.stdout + r.stderr
    for i in range(4):
        assert open(out + ".%d" % i, "rb").read() == alone[i], i


def _read_rgb_pcd(path):
    raw = open(path, "rb").read()
    k = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    hdr = raw[:k].decode()
    assert "FIELDS x y z rgb" in hdr and "SIZE 4 4 4 4" in hdr
    n = int([ln for ln in hdr.splitlines() if ln.startswith("POINTS")][0].split()[1])
    rec = np.frombuffer(raw[k:], dtype=np.dtype([("xyz", "<f4", 3), ("rgb", "<u4")]), count=n)
    return rec["xyz"].copy(), rec["rgb"].copy()


def test_show_dump_and_timing_csv_of_the_connect_planner(engine_mod, oracle_mod, tmp_path):
    """show() (path_slicing_alg.cpp:69-80) without a viewer: PPP_SHOW_PCD=<file> receives other_cloud + cloud -- the inserted
    nodes in red, then the cloud in white with the cloud point nearest to every millimetre of every path painted blue
    (drawpath, path_dynamic_alg.cpp:330,354) --; and GenPath of the derived planner appends its microseconds to output.csv
    (path_dynamic_alg.cpp:380-388)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "connect"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("tiny_5k")
    pcd = str(tmp_path / "w.pcd")
    engine_mod.save_pcd(pcd, pts)
    out = str(tmp_path / "wp.txt")
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 6\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\nSmooth = false\n"
                    "Alignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = false\n" % out)
    dump = str(tmp_path / "show.pcd")
    for rep in range(2):
        r = subprocess.run([os.path.join(root, "examples", "connect"), pcd], env=dict(os.environ, PPP_CONFIG=str(conf), PPP_SHOW_PCD=dump),
                           capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
    assert "Toal Using Time:" in r.stdout and "Number of Point Cloud: %d" % len(pts) in r.stdout
    csv = open(str(tmp_path / "output.csv")).read().split()
    assert len(csv) == 2 and all(int(v) > 0 for v in csv)               # one line per run, appended
    xyz, rgb = _read_rgb_pcd(dump)
    e = engine_mod.Engine(0, tool_radius=6.0, walk=1); e.set_cloud(pts); S = e.gen_path()
    knots = np.concatenate([np.stack([kx, ky, kz], axis=1) for ky, kx, kz in (e.nodes(s) for s in range(S))]).astype(np.float32)
    assert len(xyz) == len(knots) + len(pts)
    assert np.array_equal(xyz[:len(knots)], knots) and np.all(rgb[:len(knots)] == 0xFF0000)
    assert np.array_equal(xyz[len(knots):], e.cloud())
    crgb = rgb[len(knots):]
    painted = np.nonzero(crgb != 0xFFFFFF)[0]
    assert np.all(crgb[painted] == 0x0000FF) and len(painted) > 0
    want = set()
    for s in range(S):
        ky = e.nodes(s)[0]
        q = np.arange(ky[0], ky[-1], 1.0)                                # dy = miny; while (dy < maxy) ...; dy += 1
        rc, p = e.eval_spline(s, q)
        want.update(e.nearest(p.astype(np.float32)).tolist())
    assert set(painted.tolist()) == want


def test_show_dump_paints_the_boundary_curves_of_the_dynamic_adjustment(engine_mod, oracle_mod, tmp_path):
    """thread_worker paints every boundary curve green before it adjusts a slice against it (drawpath(*boundary, 0,255,0),
    path_dynamic_alg.cpp:320-322), then the adjusted path blue (:330): the dump of examples/connect with
    Dynamic_adjustment = true holds both, painted in the chains' order (a later curve recolours an earlier one where they
    share a cloud point).  The boundaries themselves (ppp_get_boundary) are the oracle's, knot for knot."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "connect"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("small_40k")
    pcd = str(tmp_path / "w.pcd")
    engine_mod.save_pcd(pcd, pts)
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 6\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\nSmooth = false\n"
                    "Alignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = true\n" % str(tmp_path / "wp.txt"))
    dump = str(tmp_path / "show.pcd")
    r = subprocess.run([os.path.join(root, "examples", "connect"), pcd], env=dict(os.environ, PPP_CONFIG=str(conf), PPP_SHOW_PCD=dump),
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    xyz, rgb = _read_rgb_pcd(dump)
    kw = dict(tool_radius=6.0, walk=1, dynamic_adjustment=1)
    e = engine_mod.Engine(0, **kw); e.set_cloud(pts); S = e.gen_path()
    o = oracle_mod.Oracle(pts, **kw); assert o.gen_path() == S
    nk = sum(len(e.nodes(s)[0]) for s in range(S))
    assert len(xyz) == nk + len(pts)
    crgb = rgb[nk:]
    want = np.full(len(pts), 0xFFFFFF, np.uint32)
    steps, nb = [], 0
    for s in range(S):
        by, bx, bz, step = e.boundary(s)
        oy, ox, oz = o.boundary(s)
        assert np.array_equal(by, oy) and np.array_equal(bx, ox) and np.array_equal(bz, oz), s
        steps.append((step, s))
    assert sorted(st for st, _ in steps)[:2] == [-1, 0]                  # one start slice, the chains go out from it
    for step, s in sorted(steps):
        by, bx, bz, _ = e.boundary(s)
        if len(by) >= 3:
            nb += 1
            rc, p = engine_mod.Spline(by, bx, bz).point(np.arange(by[0], by[-1], 1.0))
            assert rc == 0
            want[e.nearest(p.astype(np.float32))] = 0x00FF00
        ky = e.nodes(s)[0]
        rc, p = e.eval_spline(s, np.arange(ky[0], ky[-1], 1.0))
        want[e.nearest(p.astype(np.float32))] = 0x0000FF
    assert nb == S - 1 and "%d boundary curves painted" % nb in r.stdout
    assert np.array_equal(crgb, want)
    assert (want == 0x00FF00).sum() > 100 and (want == 0x0000FF).sum() > 100


def test_robot_path_class_end_to_end(engine_mod, oracle_mod, tmp_path):
    """include/robot_path.h (class RobotPath, robot_path.h:58-98; behaviour of the July snapshot path_connect_ex0720.cpp): the
    three-argument constructor, single-direction float walk from min.x + Radius, +-5 trim, no first/last drop, no
    position smoothing, its own hand-eye calibration -- through examples/robot against the oracle with those parameters."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "robot"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("small_40k")
    pcd = str(tmp_path / "workpiece.pcd")
    engine_mod.save_pcd(pcd, pts, binary=True)
    out = str(tmp_path / "WayPoints.txt")
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 12\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\n"
                    "Smooth = false\nAlignment = false\nChangeRange = true\nRemoveOutlier = false\n" % out)
    r = subprocess.run([os.path.join(root, "examples", "robot"), pcd, "7.5"], env=dict(os.environ, PPP_CONFIG=str(conf)),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(out, dtype=np.float64).reshape(-1, 6)
    he = [0.792078, -0.042662, 0.6656017, -3.1531625, -0.048573, 1.609157]
    o = oracle_mod.Oracle(pts, tool_radius=7.5, pairing=0, walk=3, trim=5.0, drop_ends=0, smooth=0, handeye=he)
    S = o.gen_path(); o.get_path()
    want = o.waypoints()
    assert "waypoints: %d" % len(want) in r.stdout
    assert got.shape == want.shape and len(o.tail_index()) == S          # every slice is kept
    assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-4 + 5e-6 * np.abs(want[:, :3]).max()


def test_contour_cli_trims_five_and_uses_its_own_hand_eye(engine_mod, oracle_mod, tmp_path):
    """include/contour_alg.h driven like src/contour.cpp (+ the plan): SectPath with getPath's +-5 trim
    (contour_alg.cpp:496-497) and the hand-eye calibration of contour_alg.h:37-42."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "contour"], stdout=subprocess.DEVNULL)
    pts, cfg = synth.make_config("small_40k")
    pcd = str(tmp_path / "workpiece.pcd")
    engine_mod.save_pcd(pcd, pts, binary=True)
    out = str(tmp_path / "WayPoints.txt")
    conf = tmp_path / "config.txt"
    conf.write_text("Tool_Radius = 6\npathFile = %s\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\n"
                    "Smooth = false\nAlignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = false\n" % out)
    env = dict(os.environ, PPP_CONFIG=str(conf), PPP_CONTOUR_PLAN="1")
    r = subprocess.run([os.path.join(root, "examples", "contour"), pcd], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(out, dtype=np.float64).reshape(-1, 6)
    he = [-0.858533, 0.075348, 0.672533, -3.138775, -0.0405313, -1.5707969]
    o = oracle_mod.Oracle(pts, tool_radius=6.0, walk=0, trim=5.0, handeye=he)
    o.gen_path(); o.get_path()
    want = o.waypoints()
    assert got.shape == want.shape
    assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-4 + 5e-6 * np.abs(want[:, :3]).max()
    o10 = oracle_mod.Oracle(pts, tool_radius=6.0, walk=0)
    o10.gen_path(); o10.get_path()
    assert len(want) > len(o10.waypoints())          # the shorter trim samples more of every path


def test_class_surface_in_cpp_dense_band_normals_and_spline(engine_mod, oracle_mod, tmp_path):
    """examples/api_check: the drop-in C++ classes themselves -- rangedX_index on a band of more than 4096 points (the
    header's two-call size query), estimate_normal() with the field left readable, and class Spline on caller-supplied
    knots (constructor, point, restart, copies by value, the GSL error cases)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "api_check"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(12)
    x = rng.uniform(0, 70.0, 114000); y = rng.uniform(-78, 78, 114000)      # ~6500 points per 4 mm band
    pts = (np.stack([x, y, 1500 + 6 * np.sin(x / 30) * np.cos(y / 40)], axis=1) / 1000).astype(np.float32)
    pcd = str(tmp_path / "dense.pcd")
    engine_mod.save_pcd(pcd, pts)
    r = subprocess.run([os.path.join(root, "examples", "api_check"), pcd, "35"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln and ln.split()[0] in
             ("ranged", "normals", "spline", "restart", "copy", "edom", "einval")}
    e = engine_mod.Engine(0, tool_radius=6.0, pairing=1, walk=3); e.set_cloud(pts)
    idx = e.ranged_x_index(35)
    assert len(idx) > 4096 and np.array_equal(idx, oracle_mod.Oracle(pts, tool_radius=6.0).ranged_x_index(35))
    assert [int(v) for v in lines["ranged"]] == [35, len(idx), int(idx.astype(np.int64).sum()), 1, int(idx[0]), int(idx[-1])]
    nrm = e.estimate_normals()
    ok = ~np.isnan(nrm[:, 0])
    assert int(lines["normals"][0]) == len(pts) and int(lines["normals"][1]) == int((~ok).sum())
    want = float(np.sum(nrm[ok, 2].astype(np.float64) + 0.5 * nrm[ok, 3].astype(np.float64)))
    assert abs(float(lines["normals"][2]) - want) <= 1e-6 * max(1.0, abs(want))
    n = 9
    i = np.arange(n)
    ky = -3.0 + 1.25 * i + 0.01 * i * i; kx = 100.0 + 0.5 * i * (i % 3); kz = 1500.0 - 0.75 * i + (i % 2)
    q = ky[0] + (ky[-1] - ky[0]) * np.arange(17) / 16.0; q[-1] = ky[-1]
    got = np.array([float(v) for v in lines["spline"]])
    assert got[0] == n and got[1] == ky[0] and got[2] == ky[-1]
    want = np.stack([oracle_mod.steffen(ky, kx, q)[1], q, oracle_mod.steffen(ky, kz, q)[1]], axis=1).ravel()
    assert np.array_equal(got[3:], want)
    y2 = np.array([0.0, 1.0, 2.5, 4.0]); x2 = np.array([1.0, 3.0, 2.0, 5.0]); z2 = np.array([0.0, -1.0, -1.5, 2.0])
    q2 = 0.5 * np.arange(9)
    got = np.array([float(v) for v in lines["restart"]])
    assert got[0] == 4 and got[1] == 0.0 and got[2] == 4.0
    assert np.array_equal(got[3:], np.stack([oracle_mod.steffen(y2, x2, q2)[1], oracle_mod.steffen(y2, z2, q2)[1]], axis=1).ravel())
    assert [float(v) for v in lines["copy"]] == [kx[3], kz[3]]                # the copy still holds the first fit
    assert lines["edom"] == ["1"] and lines["einval"] == ["0"]
    assert "interpolation error" in r.stderr and "at least 3 knots" in r.stderr


@pytest.mark.parametrize("walk", [0, 1, 2, 3, 4])
def test_slice_walks_on_device_match_the_reference_loops(engine_mod, oracle_mod, walk):
    """a3: the device walk (closed forms / predicated float accumulation) against the literal loops."""
    for R, x0, nx in [(6.0, 0.5, 300), (7.5, -93.7, 260), (12.0, 17.25, 400), (2.6, -200.3, 180), (15.0, 1000.2, 500)]:
        pts = synth.make_plate(nx, 24, kind="flat", seed=int(R * 10) + walk, x0_mm=x0)
        o = oracle_mod.Oracle(pts, tool_radius=R, walk=walk)
        e = engine_mod.Engine(0, tool_radius=R, walk=walk)
        e.set_cloud(pts)
        assert np.array_equal(e.slice_positions(), o.slice_positions()), (R, x0, walk)


def test_run_async_graph_replay_matches_plain_calls(engine_mod):
    pts, cfg = synth.make_config("small_40k")
    a = engine_mod.Engine(0, tool_radius=6.0); a.set_cloud(pts); a.gen_path(); a.get_path()
    want = a.waypoints().tobytes()
    b = engine_mod.Engine(0, tool_radius=6.0); b.set_cloud(pts)
    for _ in range(3):  # capture, then two replays
        b.run_async(); b.sync()
        assert b.waypoints().tobytes() == want
    b.set_params(tool_radius=9.0)          # replans: the graph must be rebuilt
    b.run_async(); b.sync()
    c = engine_mod.Engine(0, tool_radius=9.0); c.set_cloud(pts); c.gen_path(); c.get_path()
    assert b.waypoints().tobytes() == c.waypoints().tobytes()
    pts2, _ = synth.make_config("tiny_5k")
    b.set_cloud(pts2); b.run_async(); b.sync()  # new cloud: new plan, new graph
    d = engine_mod.Engine(0, tool_radius=9.0); d.set_cloud(pts2); d.gen_path(); d.get_path()
    assert b.waypoints().tobytes() == d.waypoints().tobytes()


def test_dense_cloud_overflows_lds_and_takes_the_arena_path(engine_mod, oracle_mod):
    """Maximum sizes: bands of > 4096 points do not fit the workgroup's LDS; the engine re-runs the
    slices (and slabs) that overflow on a global arena, transparently, with identical results."""
    pts = synth.make_plate(64, 2600, kind="wavy", amp=3.0, seed=31)   # 166k points, ~7000 per 4 mm band
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=8.0, path_resolution=40.0)
    assert_full_parity(engine_mod, e, o, every_slice=False)
    assert max(len(o.slice_indices(s)) for s in range(o.num_slices())) > 4096
    a = e.waypoints().tobytes()
    e.run_async(); e.sync()
    assert e.waypoints().tobytes() == a


def test_whole_cloud_normal_field(engine_mod, oracle_mod):
    """estimate_normal() as a public method: every point's PCL normal."""
    pts, cfg = synth.make_config("small_40k")
    pts = pts.copy(); pts[123] = np.nan
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    n = e.estimate_normals(); on = o.estimate_normals()
    on[123] = np.nan  # the oracle leaves a dropped point's slot at zero; PCL gives NaN for it
    keep = np.arange(len(n)) != 123
    assert np.array_equal(n[keep].view(np.uint32), on[keep].view(np.uint32))   # same neighbour order, same rounding: same bits
    nan = np.isnan(on[:, 0])
    assert np.array_equal(np.isnan(n[:, 0]), nan) and nan.sum() >= 1
    ang = np.arctan2(np.linalg.norm(np.cross(n[~nan, :3], on[~nan, :3]), axis=1), np.sum(n[~nan, :3] * on[~nan, :3], axis=1))
    assert ang.max() < 1e-4 and np.abs(n[~nan, 3] - on[~nan, 3]).max() < 1e-5


def test_duplicate_points_resolve_ties_like_the_reference(engine_mod, oracle_mod):
    """Collisions: exact coordinate duplicates give exact distance ties (lowest index wins a
    nearest-neighbour tie, highest index writes the map key last)."""
    pts, cfg = synth.make_config("tiny_5k")
    rng = np.random.default_rng(2)
    dup = rng.integers(0, len(pts), 600)
    both = np.concatenate([pts, pts[dup]])[rng.permutation(len(pts) + 600)]
    for pairing in (0, 1):
        e, o = run_pair(engine_mod, oracle_mod, both, tool_radius=6.0, pairing=pairing)
        assert_full_parity(engine_mod, e, o)


def test_points_exactly_on_a_plane_are_on_neither_side(engine_mod, oracle_mod):
    """x == plane_x: in the PassThrough band, but neither El nor Er (path_slicing_alg.cpp:180-181)."""
    pts = synth.make_plate(90, 40, kind="wavy", amp=4.0, seed=6)
    o0 = oracle_mod.Oracle(pts, tool_radius=6.0, walk=2)   # integer planes
    px = o0.slice_positions()
    mm = pts * np.float32(1000)
    for p in px[1:-1:2]:                                    # snap the nearest column onto every other plane
        col = np.abs(mm[:, 0] - p) < 0.75
        pts[col, 0] = np.float32(p) / np.float32(1000)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0, walk=2)
    assert_full_parity(engine_mod, e, o)
    on_plane = sum(int((o.points()[o.slice_indices(s), 0] == o.slice_positions()[s]).sum()) for s in range(o.num_slices()))
    assert on_plane > 50


@pytest.mark.parametrize("name", ["cfg4_2m_s256", "cfg5_10m_s1024"])
def test_large_baseline_configs(engine_mod, oracle_mod, name):
    """BASELINE.json configs 4 and 5 at full size (per-GPU share of config 4; all of config 5 on one
    GPU): counts, knots and sampled waypoints equal the oracle's, final list within 1e-4 m, plus the
    size-independent properties of the path."""
    pts, cfg = synth.make_config(name)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=cfg["tool_radius"])
    S = e.gen_path(); W = e.get_path()
    assert S == cfg["slices"] == o.gen_path() and W == o.get_path()
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_XYZ), o.waypoints_xyz())
    assert np.array_equal(e.stage(engine_mod.STAGE_WP_NN), o.waypoint_nn())
    wp, owp = e.waypoints(), o.waypoints()
    assert np.linalg.norm(wp[:, :3] - owp[:, :3], axis=1).max() <= TOL_M
    tail = e.tail_index()
    assert np.array_equal(tail, o.tail_index()) and tail[-1] == W - 1 and np.all(np.diff(tail) > 0)
    xyz = e.stage(engine_mod.STAGE_WP_XYZ)
    px = e.slice_positions()
    start = 0
    for k, t in enumerate(tail):
        seg = xyz[start:t + 1]
        assert np.all(seg[:, 0] == px[k + 1])                       # the x spline is the plane
        d = np.diff(seg[:, 1])
        assert np.all(d > 0) if k % 2 == 0 else np.all(d < 0)       # boustrophedon
        start = t + 1
    for s in range(0, S, max(1, S // 16)):
        y, x, z = e.nodes(s)
        assert np.all(np.diff(y) > 0) and len(y) >= 3               # map order, strictly increasing knots


# ---------------- dynamic adjustment ----------------
def test_area2cloud_api(engine_mod, oracle_mod):
    pts = synth.make_plate(160, 90, kind="blade", amp=25.0, seed=15)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    cloud = o.points().astype(np.float64)
    rng = np.random.default_rng(8)
    q = cloud[rng.integers(0, len(cloud), 300)] + rng.normal(0, 0.4, (300, 3))
    for key in (0, 1):
        got = e.area2cloud(q, key)
        want = np.stack([o.area2cloud(p, key) for p in q])
        nan = np.isnan(want[:, 0])
        assert np.array_equal(np.isnan(got[:, 0]), nan)
        # the normal field is bit-identical to the oracle's (distance-ordered covariance sums, correctly rounded trig in
        # computeRoots), so the principal direction and with it the extreme 0.5-degree ellipse sample are the same
        assert np.array_equal(got[~nan], want[~nan])


def test_area2cloud_with_equal_distances(engine_mod, oracle_mod):
    """Every fifth point of the cloud exists twice (same coordinates, two cloud indices): the k-NN selection meets equal
    distances in nearly every neighbourhood and must break them by cloud index, as the oracle's (distance, index) order
    does -- the ranking's second path (whole 64-bit key) next to its first (distance bits only)."""
    base = synth.make_plate(150, 80, kind="blade", amp=25.0, seed=19)
    rng = np.random.default_rng(3)
    dup = base[rng.permutation(len(base))[: len(base) // 5]]
    pts = np.concatenate([base, dup])[rng.permutation(len(base) + len(dup))].astype(np.float32)
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=6.0)
    cloud = o.points().astype(np.float64)
    q = cloud[rng.integers(0, len(cloud), 300)] + rng.normal(0, 0.3, (300, 3))
    for key in (0, 1):
        got = e.area2cloud(q, key)
        want = np.stack([o.area2cloud(p, key) for p in q])
        nan = np.isnan(want[:, 0])
        assert np.array_equal(np.isnan(got[:, 0]), nan)
        assert np.array_equal(got[~nan], want[~nan])
    assert (~nan).sum() > 100


@pytest.mark.parametrize("name,walk", [("small_40k", 1), ("small_40k", 2), ("small_40k", 3), ("cfg2_1m_s256", 1)])
def test_dynamic_adjustment_pipeline(engine_mod, oracle_mod, name, walk):
    """GenPath with Dynamic_adjustment = true (config.txt:13) for connect (walk 1), connect1 (walk 2) and
    Contact_Path_Generation of ./main (walk 3: brute pairing, 10 neighbours, inner samples only):
    every adjusted knot is a cloud point, so the knot lists must be identical.  The last case is BASELINE configs[1] at full
    size (1 M points, 256 slices: two chains of 128 steps) -- the reference's default mode on the headline workload."""
    pts, cfg = synth.make_config(name)
    kw = dict(tool_radius=cfg["tool_radius"], walk=walk, dynamic_adjustment=1)
    if walk == 3:
        kw.update(pairing=1, curvature_k=10, depth=0.005)
    o = oracle_mod.Oracle(pts, **kw)
    e = engine_mod.Engine(0, **kw)
    e.set_cloud(pts)
    So = o.gen_path(); S = e.gen_path()
    assert S == So > 10
    bad = 0
    for s in range(S):
        gy, gx, gz = e.nodes(s); oy, ox, oz = o.nodes(s)
        if len(gy) != len(oy) or not (np.array_equal(gy, oy) and np.array_equal(gx, ox) and np.array_equal(gz, oz)):
            bad += 1
    assert bad == 0
    nb = 0
    for s in range(S):                                                   # the boundary every slice was adjusted against
        by, bx, bz, step = e.boundary(s); oy, ox, oz = o.boundary(s)
        assert np.array_equal(by, oy) and np.array_equal(bx, ox) and np.array_equal(bz, oz), s
        nb += len(by) >= 3
    assert nb >= S - 2
    Wo = o.get_path(); W = e.get_path()
    assert W == Wo
    wp, owp = e.waypoints(), o.waypoints()
    assert np.linalg.norm(wp[:, :3] - owp[:, :3], axis=1).max() <= TOL_M
    # and it differs from the equal-spacing path
    e2 = engine_mod.Engine(0, tool_radius=cfg["tool_radius"], walk=walk); e2.set_cloud(pts); e2.gen_path(); e2.get_path()
    a, b = e2.stage(engine_mod.STAGE_WP_XYZ)[:, 0], e.stage(engine_mod.STAGE_WP_XYZ)[:, 0]
    assert a.shape != b.shape or np.abs(a - b).max() > 0.05   # v1 drops the end samples, so its paths get shorter
    by, bx, bz, step = e2.boundary(1)                         # no adjustment, no boundaries
    assert len(by) == 0 and step == -1
    with pytest.raises(engine_mod.PPPError):
        e.boundary(S)


def test_dense_slabs_under_dynamic_adjustment_wait_for_the_arena_pass(engine_mod, oracle_mod):
    """A square plate lying diagonally in x/y: the x-slabs through its middle hold twice the mean population -- more than a slab's
    LDS sort takes --, so the first index of the pass is incomplete and the engine repeats the pass with the arena sort.  With the
    dynamic adjustment the whole-cloud normals and the Area2Cloud searches run behind that first index in the stream: until the arena
    pass such a slab must hold valid points, an empty y-bucket row and open x bounds (a slab that kept what an earlier plan had left
    there sent the normals' stores outside their buffer: found by a randomised case).  The handle plans a larger cloud first, so
    that its index buffers are full of another plan's data.  Knots and list against the oracle."""
    big, _ = synth.make_config("small_40k")
    grid = synth.make_plate(200, 200, kind="dome", amp=12.0, seed=5)
    c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)
    ctr = grid[:, :2].mean(axis=0)
    xy = (grid[:, :2] - ctr) @ np.array([[c, -s_], [s_, c]], np.float64).T
    pts = np.ascontiguousarray(np.column_stack([xy[:, 0] + 0.35, xy[:, 1], grid[:, 2]]).astype(np.float32))
    for kw in (dict(tool_radius=7.5, walk=3, pairing=1, dynamic_adjustment=1, curvature_k=10, depth=0.005), dict(tool_radius=6.0, walk=1, dynamic_adjustment=1)):
        e = engine_mod.Engine(0, **kw)
        e.set_cloud(big)
        try:
            e.gen_path(); e.get_path()
        except engine_mod.PPPError:
            pass
        o = oracle_mod.Oracle(pts, **kw)
        e.set_cloud(pts)
        So = o.gen_path()
        try:
            S = e.gen_path()
        except engine_mod.PPPError:
            assert So < 0 and e.failed_slice() == -(So + 1)      # (the tips of the diamond: both sides give up at the same slice)
            continue
        assert S == So > 10
        for s in range(S):
            assert all(np.array_equal(a, b) for a, b in zip(e.nodes(s), o.nodes(s))), s
        W, Wo = e.get_path(), o.get_path()
        assert W == Wo and np.nan_to_num(np.linalg.norm(e.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1)).max() <= TOL_M
        e.close()


def test_dynamic_fit_paths_agree():
    """The fits of the chain (compute_boundary's std::map + end knots, dynamic_adjust_path's map + Spline::restart) take one
    of three paths: samples already in knot order, equal y to merge, unsorted.  The synthetic clouds mostly take the
    first; test builds that force the second and the third on every fit must reproduce the product build's knots and
    waypoints bit for bit.  Likewise Area2Cloud's windowed ellipse extremum (63 samples around the analytic peak when
    the error bound allows): a build that evaluates all 721 samples as well, every time, and raises a device error on
    any difference must run clean (tools/dyn_variants_check.py; the variants are compiled here, ~2 min, when they are
    missing or stale)."""
    import os, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "polishpathplanning_amd", "csrc")
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc to build the test variants with")
    jobs = [subprocess.Popen(["make", "-C", src, "variant", "NAME=%s" % name, "DEFS=%s" % defs], stdout=subprocess.DEVNULL)
            for name, defs in (("fitcount", "-DDYN_FIT_FORCE=1"), ("fitsort", "-DDYN_FIT_FORCE=3"), ("ellcheck", "-DDYN_ELL_CHECK"))]
    for j in jobs:
        assert j.wait(timeout=900) == 0, j.args
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dyn_variants_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "the three paths agree" in r.stdout, r.stdout + r.stderr


# ---------------- slice-range sharding of one cloud (SURVEY.md 8e case ii) ----------------
def _sharded_path(engine_mod, pts, world, **kw):
    """What bench.py --mode slices does with `world` GPUs, on one GPU: one handle per slice range, blocks
    concatenated in rank order, the list finished once on handle 0."""
    from polishpathplanning_amd.robot_path import slice_ranges
    probe = engine_mod.Engine(0, **kw); probe.set_cloud(pts)
    S = len(probe.slice_positions())
    W_cap = probe.gen_path() and probe.get_path()
    gathered = _DeviceBuffer(max(W_cap, 1) * 24)         # stands in for the RCCL receive buffer on rank 0
    engines, counts, W = [], None, 0
    for b, e in slice_ranges(S, world):
        if b == e:                                       # more ranks than slices: this rank contributes nothing
            continue
        eng = engine_mod.Engine(0, slice_begin=b, slice_end=e, **kw)
        eng.set_cloud(pts)
        eng.gen_path(); w = eng.get_path()
        c = eng.waypoint_counts()
        assert c.sum() == w
        k0 = 1 if kw.get("drop_ends", 1) else 0
        assert all(c[k] == 0 for k in range(len(c)) if not (b <= k + k0 < e))
        assert eng.copy_stage_to_device(engine_mod.STAGE_WP_PRESMOOTH, gathered.ptr + 24 * W, W_cap - W) == w
        W += w
        counts = c if counts is None else counts + c
        engines.append(eng)
    fin = engines[0]
    fin.finish_path_async(gathered.ptr, W, counts)
    fin.sync()
    return fin, gathered.to_host(W * 6), counts, engines


class _DeviceBuffer:
    """hipMalloc'ed bytes through the HIP runtime the engine already loaded (no torch in these tests)."""

    def __init__(self, nbytes):
        from polishpathplanning_amd.hipbuf import DeviceBuffer
        self._b = DeviceBuffer(nbytes)
        self.ptr, self.nbytes = self._b.ptr, nbytes

    def to_host(self, nfloats):
        return self._b.to_host(nfloats).reshape(-1, 6)


def test_finish_on_a_fresh_handle_that_never_ran_a_pass(engine_mod):
    """ppp_finish_path_async on a handle that was created, given the cloud and nothing else (rank 0 of --mode slices may do
    exactly that): the meta block of such a handle has never been written by a pass, so nothing in it may ask for a re-run
    (a stale win_flag made ppp_sync repeat the plan on the slab path over the list just finished) and the getters must
    report the finished list."""
    pts, cfg = synth.make_config("small_40k")
    one = engine_mod.Engine(0, tool_radius=6.0); one.set_cloud(pts); one.gen_path(); W = one.get_path()
    pre = one.stage(engine_mod.STAGE_WP_PRESMOOTH)
    counts = one.waypoint_counts()
    src = _DeviceBuffer(W * 24)
    assert one.copy_stage_to_device(engine_mod.STAGE_WP_PRESMOOTH, src.ptr, W) == W
    for fast in (True, False):
        fresh = engine_mod.Engine(0, tool_radius=6.0, fast_path=fast)
        fresh.set_cloud(pts)
        fresh.finish_path_async(src.ptr, W, counts)
        fresh.sync()
        assert fresh.num_waypoints() == W
        assert fresh.waypoints().tobytes() == one.waypoints().tobytes()
        assert np.array_equal(fresh.tail_index(), one.tail_index())
        assert np.array_equal(fresh.stage(engine_mod.STAGE_WP_PRESMOOTH), pre)
        fresh.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slice_range_sharding_is_bit_identical_to_one_handle(engine_mod, world):
    pts, cfg = synth.make_config("small_40k")
    one = engine_mod.Engine(0, tool_radius=6.0); one.set_cloud(pts); one.gen_path(); W = one.get_path()
    fin, pre, counts, engines = _sharded_path(engine_mod, pts, world, tool_radius=6.0)
    assert pre.shape[0] == W and np.array_equal(counts, one.waypoint_counts())
    assert np.array_equal(pre, one.stage(engine_mod.STAGE_WP_PRESMOOTH))
    assert fin.num_waypoints() == W
    assert fin.waypoints().tobytes() == one.waypoints().tobytes()
    assert np.array_equal(fin.tail_index(), one.tail_index())
    assert fin.smooth_sweeps() == one.smooth_sweeps()
    # a range handle has no final list of its own
    with pytest.raises(engine_mod.PPPError):
        engines[1].waypoints()


def test_slice_range_handles_fed_with_their_part_only(engine_mod):
    """SURVEY.md 8e case ii, pre-partitioned by x: a rank that never holds the whole cloud.  ppp_range_interval tells which x
    interval a slice range needs; ppp_set_cloud_part takes just those points (+ their cloud indices) and the whole cloud's
    bounds and count -- what an all-reduce of 3 minima, 3 maxima and a count gives the ranks -- and the handle plans exactly
    what a whole-cloud range handle plans: same blocks, same indices, bit-identical final list."""
    from polishpathplanning_amd.robot_path import slice_ranges
    pts, cfg = synth.make_config("small_40k")
    pts = pts.copy(); pts[17] = np.nan                               # a dropped point: not part of bounds or count
    one = engine_mod.Engine(0, tool_radius=6.0); one.set_cloud(pts); S = one.gen_path(); W = one.get_path()
    scaled = (pts * np.float32(1000)).astype(np.float32)             # the planner's units (k_ingest: x * 1000 in float)
    fin = np.isfinite(scaled).all(axis=1)
    mn, mx, nvalid = scaled[fin].min(axis=0), scaled[fin].max(axis=0), int(fin.sum())
    world = 3
    gathered = _DeviceBuffer(max(W, 1) * 24)
    counts, off, engines = None, 0, []
    for b, e in slice_ranges(S, world):
        whole = engine_mod.Engine(0, tool_radius=6.0, slice_begin=b, slice_end=e); whole.set_cloud(pts); whole.gen_path(); whole.get_path()
        g = engine_mod.Engine(0, tool_radius=6.0, slice_begin=b, slice_end=e)
        lo, hi, S2 = g.range_interval(mn[0], mx[0])
        assert S2 == S
        keep = np.nonzero((scaled[:, 0] >= lo) & (scaled[:, 0] <= hi))[0]
        assert 0 < len(keep) < len(pts)
        g.set_cloud_part(pts[keep], keep, mn, mx, nvalid, lo, hi)
        assert g.gen_path() == S
        w = g.get_path()
        assert w == whole.num_waypoints()
        for st in (engine_mod.STAGE_WP_XYZ, engine_mod.STAGE_WP_NN, engine_mod.STAGE_WP_PRESMOOTH):
            assert np.array_equal(g.stage(st), whole.stage(st), equal_nan=True), (b, e, st)
        s_mid = (b + e) // 2
        assert np.array_equal(g.slice_indices(s_mid), one.slice_indices(s_mid))         # cloud indices, not part positions
        assert all(np.array_equal(a, c) for a, c in zip(g.nodes(s_mid), one.nodes(s_mid)))
        with pytest.raises(engine_mod.PPPError):
            g.estimate_normals()                                      # indexed by the whole cloud: refused on a part
        assert g.copy_stage_to_device(engine_mod.STAGE_WP_PRESMOOTH, gathered.ptr + 24 * off, W - off) == w
        off += w
        c = g.waypoint_counts()
        counts = c if counts is None else counts + c
        engines.append(g)
    engines[0].finish_path_async(gathered.ptr, off, counts); engines[0].sync()
    assert off == W and engines[0].waypoints().tobytes() == one.waypoints().tobytes()
    # a part that does not cover the range's interval, or no slice range at all, is refused
    g = engine_mod.Engine(0, tool_radius=6.0, slice_begin=2, slice_end=9)
    lo, hi, _ = g.range_interval(mn[0], mx[0])
    keep = np.nonzero((scaled[:, 0] >= lo + 30) & (scaled[:, 0] <= hi))[0]
    with pytest.raises(engine_mod.PPPError):
        g.set_cloud_part(pts[keep], keep, mn, mx, nvalid, lo + 30, hi)
    g2 = engine_mod.Engine(0, tool_radius=6.0)
    with pytest.raises(engine_mod.PPPError):
        g2.set_cloud_part(pts, None, mn, mx, nvalid, mn[0], mx[0])


def test_slice_range_sharding_cfg5_in_8_parts_parity_with_the_oracle(engine_mod, oracle_mod):
    """BASELINE configs[4] as bench.py --mode slices shards it over 8 GPUs, on one: the real 10 M-point / 1024-slice cloud, eight
    handles that each hold ONLY the points of their slice range's x interval (ppp_range_interval + ppp_set_cloud_part, with the
    whole cloud's bounds and count), the pre-smoothing blocks concatenated in rank order and finished once -- against the oracle."""
    from polishpathplanning_amd.robot_path import slice_ranges
    pts, cfg = synth.make_config("cfg5_10m_s1024")
    R = cfg["tool_radius"]
    o = oracle_mod.Oracle(pts, tool_radius=R)
    So = o.gen_path(); Wo = o.get_path()
    assert So == cfg["slices"]
    scaled = (pts * np.float32(1000)).astype(np.float32)
    mn, mx, nvalid = scaled.min(axis=0), scaled.max(axis=0), len(pts)
    gathered = _DeviceBuffer(Wo * 24)
    counts, off, first = None, 0, None
    for b, e in slice_ranges(So, 8):
        g = engine_mod.Engine(0, tool_radius=R, slice_begin=b, slice_end=e)
        lo, hi, S2 = g.range_interval(mn[0], mx[0])
        assert S2 == So
        keep = np.nonzero((scaled[:, 0] >= lo) & (scaled[:, 0] <= hi))[0]
        assert len(keep) < len(pts) // 6                                  # an eighth of the cloud plus the margins
        g.set_cloud_part(pts[keep], keep, mn, mx, nvalid, lo, hi)
        assert g.gen_path() == So
        w = g.get_path()
        assert g.copy_stage_to_device(engine_mod.STAGE_WP_PRESMOOTH, gathered.ptr + 24 * off, Wo - off) == w
        off += w
        c = g.waypoint_counts()
        counts = c if counts is None else counts + c
        if first is None:
            first = g
        else:
            g.close()
    assert off == Wo
    first.finish_path_async(gathered.ptr, off, counts); first.sync()
    wp, owp = first.waypoints(), o.waypoints()
    assert np.linalg.norm(wp[:, :3] - owp[:, :3], axis=1).max() <= TOL_M
    d = np.abs(wp[:, 3:] - owp[:, 3:])
    assert np.minimum(d, np.abs(d - 2 * np.pi)).max() <= TOL_RAD
    assert np.array_equal(first.tail_index(), o.tail_index())


def test_slice_range_sharding_cfg3_through_8_handles_parity_with_the_oracle(engine_mod, oracle_mod):
    """cfg 3 geometry (250 k points, 128 slices) through 8 whole-cloud range handles, against the oracle's list."""
    pts, cfg = synth.make_config("cfg3_250k_s128")
    fin, pre, counts, _ = _sharded_path(engine_mod, pts, 8, tool_radius=cfg["tool_radius"])
    o = oracle_mod.Oracle(pts, tool_radius=cfg["tool_radius"])
    o.gen_path(); Wo = o.get_path()
    assert fin.num_waypoints() == Wo
    d = np.linalg.norm(fin.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1)
    assert d.max() <= TOL_M
    assert np.array_equal(fin.tail_index(), o.tail_index())


def test_slice_range_margin_too_small_is_reported(engine_mod):
    pts, cfg = synth.make_config("small_40k")
    e = engine_mod.Engine(0, tool_radius=6.0, slice_begin=5, slice_end=9, range_margin=5.0, normal_radius=2.5)
    e.set_cloud(pts)
    e.gen_path(); e.get_path()      # 5 mm cover the NN ball and the 2.5 mm normal neighbourhood of a waypoint on its plane
    with pytest.raises(engine_mod.PPPError):
        engine_mod.Engine(0, tool_radius=6.0, slice_begin=5, slice_end=9, range_margin=1.0)  # < 2 x normal radius
    with pytest.raises(engine_mod.PPPError):
        engine_mod.Engine(0, tool_radius=6.0, slice_begin=5, slice_end=9, dynamic_adjustment=1)


def test_run_batch_graph_matches_single_handles(engine_mod):
    """ppp_run_batch_async: several workpieces as one hipGraph with a branch each, lists landing in one device buffer."""
    # the last member has slices shorter than RPYres + 1 waypoints: the in-order B.6 path of the emitting launch (last tile to arrive), which copies
    # its list into the batch buffer from one workgroup
    kinds = [("small_40k", 1, {}), ("tiny_5k", 2, {}), ("small_40k", 3, {}), ("tiny_5k", 4, {}), ("small_40k", 5, {}),
             ("tiny_5k", 6, dict(path_resolution=9.0))]
    clouds = [synth.make_config(n, seed=s)[0] for n, s, _ in kinds]
    want = []
    for pts, (_, _, kw) in zip(clouds, kinds):
        e = engine_mod.Engine(0, tool_radius=6.0, **kw); e.set_cloud(pts); e.gen_path(); e.get_path()
        want.append(e.waypoints())
    assert e.waypoint_counts().max() <= 7
    engines = []
    for pts, (_, _, kw) in zip(clouds, kinds):
        e = engine_mod.Engine(0, tool_radius=6.0, **kw); e.set_cloud(pts); engines.append(e)
    ws = [len(w) for w in want]
    offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
    buf = _DeviceBuffer(sum(ws) * 24)
    for _ in range(3):                                   # capture, then two replays
        engine_mod.run_batch_async(engines, buf.ptr, offs, ws)
        engine_mod.sync_batch(engines)
    got = buf.to_host(sum(ws) * 6)
    assert got.tobytes() == np.concatenate(want).tobytes()
    for e, w in zip(engines, want):
        assert e.waypoints().tobytes() == w.tobytes()
    # double buffering: two destinations in turn (each keeps its own cached graph)
    buf2 = _DeviceBuffer(sum(ws) * 24)
    for k in range(4):
        engine_mod.run_batch_async(engines, (buf if k % 2 == 0 else buf2).ptr, offs, ws)
        engine_mod.sync_batch(engines)
    assert buf2.to_host(sum(ws) * 6).tobytes() == np.concatenate(want).tobytes()
    # a destination slot that is too small is that handle's error, not silent truncation
    caps = list(ws); caps[2] -= 1
    engine_mod.run_batch_async(engines, buf.ptr, offs, caps)
    with pytest.raises(engine_mod.PPPError):
        engine_mod.sync_batch(engines)
    # without a destination, and after a parameter change of one member (its plan epoch changes)
    engines[1].set_params(path_resolution=5.0)
    engine_mod.run_batch_async(engines)
    engine_mod.sync_batch(engines)
    assert engines[1].num_waypoints() > ws[1] and engines[0].waypoints().tobytes() == want[0].tobytes()


def test_batch_of_64_cfg3_workpieces_full_size(engine_mod, oracle_mod):
    """BASELINE config 3 as stated: 64 distinct 250 k-point workpieces (own seed and dome amplitude each) planned by ONE
    ppp_run_batch_async call -- one launch per stage over all members -- and every member checked against the oracle."""
    count = 64
    rng = np.random.default_rng(3)
    amps = rng.uniform(10.0, 40.0, count)
    clouds = [synth.make_config("cfg3_250k_s128", seed=300 + i, amp=float(amps[i]))[0] for i in range(count)]
    engines = []
    for pts in clouds:
        e = engine_mod.Engine(0, tool_radius=6.0); e.set_cloud(pts); engines.append(e)
    oracles, ws = [], []
    for pts in clouds:
        o = oracle_mod.Oracle(pts, tool_radius=6.0); assert o.gen_path() == 128; ws.append(o.get_path()); oracles.append(o)
    offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
    buf = _DeviceBuffer(sum(ws) * 24)
    for _ in range(3):                                   # capture, then two replays
        engine_mod.run_batch_async(engines, buf.ptr, offs, ws)
        engine_mod.sync_batch(engines)
    got = buf.to_host(sum(ws) * 6)
    for i, (e, o) in enumerate(zip(engines, oracles)):
        assert e.num_slices() == 128 and e.num_waypoints() == ws[i], i
        assert np.array_equal(e.tail_index(), o.tail_index()), i
        for s in rng.choice(128, 6, replace=False):
            gy, gx, gz = e.nodes(int(s)); oy, ox, oz = o.nodes(int(s))
            assert np.array_equal(gy, oy) and np.array_equal(gx, ox) and np.array_equal(gz, oz), (i, s)
        wp, owp = e.waypoints(), o.waypoints()
        assert got[offs[i]: offs[i] + ws[i]].tobytes() == wp.tobytes(), i
        assert np.linalg.norm(wp[:, :3] - owp[:, :3], axis=1).max() <= TOL_M, i
        d = np.abs(wp[:, 3:] - owp[:, 3:])
        assert np.minimum(d, np.abs(d - 2 * np.pi)).max() <= TOL_RAD, i
    # the batched launches and a handle's own launch sequence give the same bytes
    solo = engine_mod.Engine(0, tool_radius=6.0); solo.set_cloud(clouds[17]); solo.gen_path(); solo.get_path()
    assert solo.waypoints().tobytes() == engines[17].waypoints().tobytes()


def test_batched_launches_random_members_match_single_handles(engine_mod):
    """Batches of workpieces of very different sizes, shapes, walks and resolutions through the batched launches (one launch
    per stage, blockIdx.y = member, every member with its own grid sizes and LDS carving): every list must carry the bytes
    the member's own launch sequence gives."""
    rng = np.random.default_rng(77)
    for rnd in range(3):
        specs = []
        for _ in range(int(rng.integers(5, 12))):
            nx, ny = int(rng.integers(60, 700)), int(rng.integers(30, 260))
            kind = ["wavy", "dome", "blade", "flat"][int(rng.integers(0, 4))]
            kw = dict(tool_radius=float(rng.choice([4.0, 6.0, 7.5, 12.0])), walk=int(rng.choice([0, 1, 2, 3])),
                      path_resolution=float(rng.choice([3.0, 7.0, 11.0])), rpy_resolution=float(rng.choice([0.0, 3.0, 7.0])),
                      trim=float(rng.choice([5.0, 10.0])), drop_ends=int(rng.integers(0, 2)), smooth=int(rng.integers(0, 2)))
            specs.append((synth.make_plate(nx, ny, kind, float(rng.uniform(2, 30)), seed=int(rng.integers(1, 10 ** 6))), kw))
        want, engines = [], []
        for pts, kw in specs:
            f = engine_mod.Engine(0, **kw); f.set_cloud(pts)
            try:
                f.gen_path(); f.get_path(); want.append(f.waypoints())
            except engine_mod.PPPError as ex:
                want.append(ex.code)
            e = engine_mod.Engine(0, **kw); e.set_cloud(pts); engines.append(e)
        ws = [0 if isinstance(w, int) else len(w) for w in want]
        offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
        buf = _DeviceBuffer(max(sum(ws), 1) * 24)
        for _ in range(2):                               # capture + replay
            engine_mod.run_batch_async(engines, buf.ptr, offs, [max(w, 1) for w in ws])
            try:
                engine_mod.sync_batch(engines)
            except engine_mod.PPPError:
                assert any(isinstance(w, int) for w in want)
            for i, (e, w) in enumerate(zip(engines, want)):
                if isinstance(w, int):
                    with pytest.raises(engine_mod.PPPError):
                        e.waypoints()
                else:
                    e.sync()
                    assert e.waypoints().tobytes() == w.tobytes(), (rnd, i)
            got = buf.to_host(max(sum(ws), 1) * 6)[:sum(ws)]
            good = [w for w in want if not isinstance(w, int)]
            if good:
                assert got.tobytes() == np.concatenate(good).tobytes()


def test_batch_member_that_overflows_lds_lands_in_the_batch_buffer(engine_mod, oracle_mod):
    """A member whose bands do not fit the LDS fast path is re-planned with the arena passes by ppp_sync_batch: the
    re-planned list must also reach that member's rows of the batch destination (the RCCL send buffer)."""
    small = synth.make_config("small_40k", seed=42)[0]
    # the same plate with a 20 mm strip ten times denser: the plan (sized from the mean density) gives bands 1024 LDS slots
    # and slabs 2048, the strip's bands hold ~2500 points and its slabs ~6000
    rng = np.random.default_rng(41)
    x = rng.uniform(290.0, 310.0, 12000); y = rng.uniform(-70.0, 70.0, 12000)
    z = 20.0 * np.sin(x / 600.0) * np.cos(y / 300.0) + 1500.0
    dense = np.concatenate([small, (np.stack([x, y, z], axis=1) / 1000).astype(np.float32)])
    probe = engine_mod.Engine(0, tool_radius=6.0); probe.set_cloud(dense); S = probe.gen_path()
    assert max(len(probe.slice_indices(s)) for s in range(S)) > 2048
    want = []
    for pts in (small, dense, small):
        f = engine_mod.Engine(0, tool_radius=6.0); f.set_cloud(pts); f.gen_path(); f.get_path(); want.append(f.waypoints())
    engines = []
    for pts in (small, dense, small):
        e = engine_mod.Engine(0, tool_radius=6.0); e.set_cloud(pts); engines.append(e)
    ws = [len(w) for w in want]
    offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
    buf = _DeviceBuffer(sum(ws) * 24)
    for _ in range(3):      # first call: batched launches + the re-run of member 1; then its arena plan keeps the batch on the branch graph
        engine_mod.run_batch_async(engines, buf.ptr, offs, ws)
        engine_mod.sync_batch(engines)
        assert buf.to_host(sum(ws) * 6).tobytes() == np.concatenate(want).tobytes()


def test_randomised_sweep_against_the_oracle(engine_mod, oracle_mod):
    """tests/tools/fuzz_parity.py, 208 seeded cases (40 standard, 60 tiny, 40 with preprocessing, 60 with odd parameters, 8 large): random shape / radius / walk / pairing / dynamic adjustment / duplicates /
    non-finite points; knots bit-exact, waypoints <= 1e-4 m, identical failing slice where the reference would abort,
    slice-range sharding byte-identical to the single handle."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tests", "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2024)
    bad = []
    for i in range(40):
        res, desc = fz.one_case(rng, i)
        if res is not None and not res.startswith("both fail"):
            bad.append((desc, res))
    os.environ["PPP_FUZZ_TINY"] = "1"     # 60 clouds of a few hundred points: one or two slices, empty sides, NaN ellipses
    try:
        rng = np.random.default_rng(51)
        for i in range(60):
            res, desc = fz.one_case(rng, i)
            if res is not None and not res.startswith("both fail"):
                bad.append((desc, res))
    finally:
        del os.environ["PPP_FUZZ_TINY"]
    os.environ["PPP_FUZZ_PRE"] = "1"  # and 40 with the constructors' preprocessing (voxel grid, MLS, alignment of a tilted plate) in front
    try:
        rng = np.random.default_rng(41)
        for i in range(40):
            res, desc = fz.one_case(rng, i)
            if res is not None and not res.startswith("both fail"):
                bad.append((desc, res))
    finally:
        del os.environ["PPP_FUZZ_PRE"]
    os.environ["PPP_FUZZ_ODD"] = "1"  # and 60 with unusual but legal parameters: tool steps of 2 .. 6 mm (overlapping bands: slab path) up to 80 mm,
    try:                              # resolutions of 0.5 .. 30 mm, RPY resolutions around the 2.0 switch, trims of 0 .. 20
        rng = np.random.default_rng(61)
        for i in range(60):
            res, desc = fz.one_case(rng, i)
            if res is not None and not res.startswith("both fail"):
                bad.append((desc, res))
    finally:
        del os.environ["PPP_FUZZ_ODD"]
    rng = np.random.default_rng(31)   # and 8 clouds of 0.2 .. 1.5 M points (long dynamic chains, many slabs)
    for i in range(8):
        res, desc = fz.one_case(rng, i, big=True)
        if res is not None and not res.startswith("both fail"):
            bad.append((desc, res))
    assert not bad, bad


@pytest.mark.parametrize("name", ["small_40k", "tiny_5k"])
def test_remove_outlier_matches_the_oracle(engine_mod, oracle_mod, name):
    """SectPath::remove_outlier (RemoveOutlier = true): same threshold, identical filtered cloud, same path afterwards."""
    pts, cfg = synth.make_config(name)
    rng = np.random.default_rng(4)
    out = pts[rng.integers(0, len(pts), 60)].copy()
    out[:, 2] += rng.uniform(0.004, 0.03, 60).astype(np.float32)
    pts = np.concatenate([pts, out])[rng.permutation(len(pts) + 60)]
    pts[7] = np.nan
    o = oracle_mod.Oracle(pts, tool_radius=6.0)
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    n_o, thr_o, _ = o.remove_outlier(50, 1.0)
    n_e, thr_e = e.remove_outlier(50, 1.0)
    assert n_e == n_o < len(pts) and abs(thr_e - thr_o) <= 1e-9 * thr_o
    assert np.array_equal(np.nan_to_num(e.cloud()), np.nan_to_num(o.points()))
    assert_full_parity(engine_mod, e, o)


@pytest.mark.parametrize("leaf", [(0.1, 1.0, 1.0), (3.0, 3.0, 3.0), (7.5, 2.0, 50.0), (1e-3, 1e-3, 1e-3)])
def test_voxel_down_matches_the_oracle(engine_mod, oracle_mod, leaf):
    """path_generater::voxel_down (pcl::VoxelGrid): same voxels in the same order, bit-identical centroids (both sides add
    in ascending point index), same behaviour on index overflow, same path afterwards."""
    pts, cfg = synth.make_config("small_40k")
    pts = pts.copy()
    pts[7] = np.nan
    o = oracle_mod.Oracle(pts, tool_radius=6.0)
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    n_o, ov_o = o.voxel_down(*leaf)
    n_e, ov_e = e.voxel_down(*leaf)
    assert (n_e, ov_e) == (n_o, ov_o)
    assert ov_o == (leaf[0] < 0.01)
    assert np.array_equal(np.nan_to_num(e.cloud()), np.nan_to_num(o.points()))
    if leaf == (0.1, 1.0, 1.0):      # main.cpp:25's leaf: the plan still has points in every band
        assert_full_parity(engine_mod, e, o)


@pytest.mark.parametrize("order", [3, 2, 1])
def test_mls_smooth_matches_the_oracle(engine_mod, oracle_mod, order):
    """SectPath::smooth (Smooth = true; pcl::MovingLeastSquares order 3 radius 15): the same points survive, coordinates
    equal to the last float bit but for rare 1-ulp cases (f64 sums grouped differently), and the plan on the smoothed
    cloud agrees within the waypoint tolerance."""
    pts, cfg = synth.make_config("small_40k")
    rng = np.random.default_rng(8)
    pts = pts.copy()
    pts[:, 2] += rng.normal(0, 0.2e-3, len(pts)).astype(np.float32)
    pts[9] = np.nan
    far = pts[:3].copy(); far[:, 2] += 0.5                    # three isolated points, 0.5 m above: fewer than 3 neighbours
    far[:, 0] += np.array([0.0, 0.2, 0.4], np.float32)
    pts = np.concatenate([pts, far])
    o = oracle_mod.Oracle(pts, tool_radius=6.0)
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    n_o = o.smooth_mls(15.0, order)
    n_e = e.smooth_mls(15.0, order)
    assert n_e == n_o == len(pts) - 4
    A, Bc = e.cloud(), o.points()
    diff = np.abs(A.astype(np.float64) - Bc.astype(np.float64))
    assert diff.max() <= 1.3e-4                                # one float ulp at |x| < 1024 mm
    assert (A != Bc).any(axis=1).mean() < 1e-3
    if order == 3:
        So = o.gen_path(); S = e.gen_path()
        assert S == So
        o.get_path(); e.get_path()
        wo, we = o.waypoints(), e.waypoints()
        assert we.shape == wo.shape
        assert np.abs(we[:, :3] - wo[:, :3]).max() < 1e-4


def _rotated_plate(t, name="small_40k"):
    """the synthetic plate in a tilted sensor frame; draw t of a fixed sequence of rotations"""
    pts, cfg = synth.make_config(name)
    rng = np.random.default_rng(1)
    for _ in range(t + 1):
        ax, ay, az = rng.uniform(-0.6, 0.6, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (pts.astype(np.float64) @ (Rz @ Ry @ Rx).T + np.array([0.3, -0.2, 0.8])).astype(np.float32)


@pytest.mark.parametrize("t", [7, 11, 0, 2])
def test_trans2center_matches_the_oracle(engine_mod, oracle_mod, t):
    """SectPath::trans2center (Alignment = true): the float running sums of pcl::compute3DCentroid / computeCovarianceMatrix
    bit for bit, the same TransAlign, the same aligned cloud; then getPath through invTransAlign with the nearest point
    and the normals taken in the cloud carried back (path_translation_alg.cpp:146-174).  Rotations 7 and 11 leave the
    thin axis third (a plan that makes sense, 11 through a reflection); 0 and 2 leave it second, where the reference's
    unsorted eigenvectors slice the sheet across its thickness -- both sides must still do the same thing."""
    pts = _rotated_plate(t)
    pts[5] = np.nan
    o = oracle_mod.Oracle(pts, tool_radius=6.0)
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    rc, To, co, covo = o.trans2center()
    Te, ce, cove = e.trans2center()
    assert rc == 0
    assert ce.tobytes() == co.tobytes() and cove.tobytes() == covo.tobytes()
    assert Te.tobytes() == To.tobytes()
    assert np.array_equal(np.nan_to_num(e.cloud()), np.nan_to_num(o.points()))
    ext = np.nanmax(o.points(), axis=0) - np.nanmin(o.points(), axis=0)
    if t in (7, 11):
        assert ext[2] < 5 < ext[1] < ext[0]
        assert_full_parity(engine_mod, e, o)
        want = e.waypoints().tobytes()
        for _ in range(3):                    # the captured graph carries the sensor-frame index of the aligned variant
            e.run_async(); e.sync()
            assert e.waypoints().tobytes() == want
        if t == 7:                            # an aligned and an unaligned handle as branches of one batch graph
            pts2, _ = synth.make_config("tiny_5k")
            e2 = engine_mod.Engine(0, tool_radius=6.0); e2.set_cloud(pts2); e2.gen_path(); e2.get_path()
            want2 = e2.waypoints().tobytes()
            for _ in range(2):
                engine_mod.run_batch_async([e, e2]); engine_mod.sync_batch([e, e2])
                assert e.waypoints().tobytes() == want and e2.waypoints().tobytes() == want2
        n_o = o.remove_outlier(50, 1.0)[0]    # constructor order: align, then remove -- the sensor-frame copy follows the cloud
        assert e.remove_outlier(50, 1.0)[0] == n_o
        assert_full_parity(engine_mod, e, o)
    else:
        So = o.gen_path()
        try:
            S = e.gen_path()
        except engine_mod.PPPError:
            S = -1
        assert (S > 0) == (So > 0)
        if So > 0:
            assert S == So
            no = o.get_path()
            try:
                e.get_path(); ne = len(e.waypoints())
            except engine_mod.PPPError:
                ne = -1
            assert (ne >= 0) == (no >= 0)
            if no > 0:
                assert ne == no and np.abs(e.waypoints()[:, :3] - o.waypoints()[:, :3]).max() < 1e-4
    with pytest.raises(engine_mod.PPPError):
        e.trans2center()                      # TransAlign would be overwritten
    e.set_cloud(pts)                          # a new cloud is unaligned again
    assert np.array_equal(np.nan_to_num(e.cloud()), np.nan_to_num(pts * np.float32(1000)))
    e.trans2center()


def test_running_float_sums_are_exact_at_a_million_points(engine_mod):
    """The wave-scan reproduction of a sequential float sum (k_seq_sum) against numpy's add.accumulate in float32:
    a million points far from the origin (many binades crossed on the way up), NaNs skipped, mixed signs."""
    pts, cfg = synth.make_config("cfg2_1m_s256")
    pts = pts.copy()
    pts[::1000] = np.nan
    pts[:, 1] -= 0.35                                          # y on both sides of zero: the running sum turns around
    e = engine_mod.Engine(0, tool_radius=6.0)
    e.set_cloud(pts)
    P = e.cloud()                                              # x1000, float
    T, c, cov = e.trans2center()
    fin = np.isfinite(P).all(axis=1)
    Q = P[fin]
    cnt = np.float32(len(Q))
    c_np = np.array([np.add.accumulate(Q[:, d], dtype=np.float32)[-1] / cnt for d in range(3)], np.float32)
    assert c.tobytes() == c_np.tobytes()
    D = Q - c_np
    want = np.zeros((3, 3), np.float32)
    for (i, j) in [(1, 1), (1, 2), (2, 2), (0, 0), (0, 1), (0, 2)]:
        prod = (D[:, j] * D[:, i]).astype(np.float32) if i == 0 else (D[:, i] * D[:, j]).astype(np.float32)
        want[i, j] = want[j, i] = np.add.accumulate(prod, dtype=np.float32)[-1]
    assert cov.tobytes() == want.tobytes()


@pytest.mark.parametrize("pairing,walk", [(1, 3), (1, 4), (0, 1)])
def test_dense_bands_beyond_lds_take_the_arena_pass(engine_mod, oracle_mod, pairing, walk):
    """A cloud so dense that every 4 mm band holds ~6500 points (more than the 4096 an LDS-resident slice kernel takes):
    the slices are parked by the first pass and planned from the arena -- for the brute-force pairing of v1
    (k_slice_brute_arena) as for the kd pairing -- with the same knots and waypoints, also when only GenPath runs."""
    rng = np.random.default_rng(12)
    x = rng.uniform(0, 70.0, 114000); y = rng.uniform(-78, 78, 114000)
    z = 1500 + 6 * np.sin(x / 30) * np.cos(y / 40)
    pts = (np.stack([x, y, z], axis=1) / 1000).astype(np.float32)
    o = oracle_mod.Oracle(pts, tool_radius=6.0, pairing=pairing, walk=walk)
    e = engine_mod.Engine(0, tool_radius=6.0, pairing=pairing, walk=walk)
    e.set_cloud(pts)
    So = o.gen_path(); S = e.gen_path()
    assert S == So >= 3
    assert max(len(e.slice_indices(s)) for s in range(S)) > 4096
    for s in range(S):
        assert all(np.array_equal(a, b) for a, b in zip(e.nodes(s), o.nodes(s))), s
    assert_full_parity(engine_mod, e, o)


def test_distinct_handles_from_concurrent_host_threads(engine_mod):
    """thread-compatible -- distinct handles may be driven from different host threads at once
    (own stream, own graph capture in thread-local mode, no globals)."""
    import threading
    kinds = [("small_40k", 11), ("tiny_5k", 12), ("small_40k", 13), ("tiny_5k", 14), ("small_40k", 15), ("tiny_5k", 16)]
    clouds = [synth.make_config(n, seed=s)[0] for n, s in kinds]
    want = []
    for pts in clouds:
        e = engine_mod.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.gen_path(); e.get_path()
        want.append(e.waypoints().tobytes())
    errors = []

    def worker(i):
        try:
            e = engine_mod.Engine(0, tool_radius=6.0)
            e.set_cloud(clouds[i])
            for rep in range(10):
                if rep % 3 == 1:
                    e.run_async(); e.sync()
                elif rep % 3 == 2:       # a parameter change drops the graph: the next run_async captures again
                    e.set_params(path_resolution=7.0); e.run_async(); e.sync()
                else:
                    e.gen_path(); e.get_path()
                if e.waypoints().tobytes() != want[i]:
                    errors.append((i, rep, "list differs"))
                e.nodes(1); e.slice_indices(2); e.tail_index(); e.stage(engine_mod.STAGE_WP_NN)   # synchronous readers
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    # ... while another thread preprocesses clouds (allocations, the library sort, a second handle made inside trans2center)
    tilted = _rotated_plate(7)
    ref = engine_mod.Engine(0, tool_radius=6.0); ref.set_cloud(tilted); ref.smooth_mls(15.0, 3); ref.trans2center(); ref.remove_outlier(50, 1.0)
    ref.gen_path(); ref.get_path()
    want_pre = ref.waypoints().tobytes()

    def preprocessor():
        try:
            for rep in range(4):
                e = engine_mod.Engine(0, tool_radius=6.0)
                e.set_cloud(tilted)
                e.smooth_mls(15.0, 3); e.trans2center(); e.remove_outlier(50, 1.0)
                e.run_async(); e.sync()
                if e.waypoints().tobytes() != want_pre:
                    errors.append(("pre", rep, "list differs"))
                e.voxel_down(0.5, 0.5, 5.0)
        except Exception as ex:  # noqa: BLE001
            errors.append(("pre", repr(ex)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(clouds))] + [threading.Thread(target=preprocessor)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors


def test_one_handle_through_many_clouds_errors_and_modes(engine_mod, oracle_mod):
    """Handle reuse: clouds of different sizes in turn, a cloud that fails in between, the dynamic adjustment switched
    on and off, a device-resident cloud, outlier removal -- after each, the result equals a fresh handle's."""
    from polishpathplanning_amd.hipbuf import DeviceBuffer
    import ctypes as C

    def fresh(pts, **kw):
        f = engine_mod.Engine(0, **kw); f.set_cloud(pts); f.gen_path(); f.get_path()
        return f.waypoints().tobytes()

    big = synth.make_config("small_40k", seed=21)[0]
    small = synth.make_config("tiny_5k", seed=22)[0]
    flat = synth.make_plate(150, 40, kind="flat", seed=23)            # dynamic adjustment fails on a plane (no curvature)
    one_sided = small.copy(); one_sided[:, 0] = np.float32(0.0405)    # every point on one plane x: no slice has two sides
    e = engine_mod.Engine(0, tool_radius=6.0)
    for step, (pts, kw) in enumerate([(big, {}), (small, {}), (big, dict(dynamic_adjustment=1)), (flat, dict(dynamic_adjustment=1)),
                                      (one_sided, dict(dynamic_adjustment=0)), (small, {}), (big, dict(path_resolution=5.0)),
                                      (np.zeros((0, 3), np.float32), {}), (big, dict(path_resolution=7.0))]):
        if kw:
            e.set_params(**kw)
        e.set_cloud(pts)
        params = dict(tool_radius=6.0, dynamic_adjustment=e.params.dynamic_adjustment, path_resolution=e.params.path_resolution)
        try:
            e.gen_path(); e.get_path()
            got = e.waypoints().tobytes()
        except engine_mod.PPPError as ex:
            got = ("error", ex.code)
        try:
            want = fresh(pts, **params)
        except engine_mod.PPPError as ex:
            want = ("error", ex.code)
        assert got == want, (step, got if isinstance(got, tuple) else len(got), want if isinstance(want, tuple) else len(want))
    # a cloud that already lives in device memory (stride 16, as a float4 array would be)
    buf = DeviceBuffer(len(big) * 16)
    host = np.zeros((len(big), 4), np.float32); host[:, :3] = big
    assert buf.hip.hipMemcpy(C.c_void_p(buf.ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) == 0
    e.set_params(dynamic_adjustment=0, path_resolution=7.0)
    e._chk(e.L.ppp_set_cloud_device(e.h, C.c_void_p(buf.ptr), len(big), 16, None))
    e.gen_path(); e.get_path()
    assert e.waypoints().tobytes() == fresh(big, tool_radius=6.0)
    # outlier removal on the reused handle, then planning
    n1, thr = e.remove_outlier(50, 1.0)
    o = oracle_mod.Oracle(big, tool_radius=6.0); n2, thr2, _ = o.remove_outlier(50, 1.0)
    assert n1 == n2
    e.gen_path(); e.get_path(); o.gen_path(); o.get_path()
    assert np.linalg.norm(e.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1).max() <= TOL_M


def test_batch_with_mixed_members_and_a_failing_one(engine_mod):
    """One batch graph over a plain handle, a dynamic-adjustment handle, a brute-pairing handle and one whose cloud
    makes the planner fail: the failure is that member's alone."""
    big = synth.make_config("small_40k", seed=31)[0]
    small = synth.make_config("tiny_5k", seed=32)[0]
    bad = small.copy(); bad[:, 0] = np.float32(0.0405)
    specs = [(big, dict()), (big, dict(dynamic_adjustment=1)), (small, dict(pairing=1, walk=3)), (bad, dict()), (small, dict(walk=2))]
    want = []
    for pts, kw in specs:
        f = engine_mod.Engine(0, tool_radius=6.0, **kw); f.set_cloud(pts)
        try:
            f.gen_path(); f.get_path(); want.append(f.waypoints())
        except engine_mod.PPPError as ex:
            want.append(ex.code)
    assert isinstance(want[3], int) and all(not isinstance(w, int) for i, w in enumerate(want) if i != 3)
    engines = []
    for pts, kw in specs:
        e = engine_mod.Engine(0, tool_radius=6.0, **kw); e.set_cloud(pts); engines.append(e)
    ws = [0 if isinstance(w, int) else len(w) for w in want]
    offs = np.concatenate([[0], np.cumsum(ws)[:-1]])
    buf = _DeviceBuffer(max(sum(ws), 1) * 24)
    for _ in range(2):
        engine_mod.run_batch_async(engines, buf.ptr, offs, [max(w, 1) for w in ws])
        with pytest.raises(engine_mod.PPPError) as ei:
            engine_mod.sync_batch(engines)
        assert "handle 3" in str(ei.value)
        for i, e in enumerate(engines):
            if i == 3:
                with pytest.raises(engine_mod.PPPError):
                    e.waypoints()
            else:
                e.sync()
                assert e.waypoints().tobytes() == want[i].tobytes()
    got = buf.to_host(sum(ws) * 6)
    assert got.tobytes() == np.concatenate([w for w in want if not isinstance(w, int)]).tobytes()


def test_gather_waypoints_single_rank_and_argument_checks(engine_mod):
    """ppp_gather_waypoints with one rank is a device copy of the list into the receive buffer (the send/recv path
    needs several GPUs and is the driver's to run); bad arguments come back as errors."""
    pts, cfg = synth.make_config("small_40k")
    e = engine_mod.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.gen_path(); W = e.get_path()
    buf = _DeviceBuffer(W * 24)
    e.gather_waypoints(0, 0, 1, 0, [W], buf.ptr)
    e.sync()
    assert buf.to_host(W * 6).tobytes() == e.waypoints().tobytes()
    with pytest.raises(engine_mod.PPPError):
        e.gather_waypoints(0, 1, 2, 0, [W, W], buf.ptr)      # two ranks but no communicator
    with pytest.raises(engine_mod.PPPError):
        e.gather_waypoints(0, 0, 1, 0, [10 ** 9], buf.ptr)   # more rows than the list can hold
    r = engine_mod.Engine(0, tool_radius=6.0, slice_begin=2, slice_end=5); r.set_cloud(pts); r.gen_path(); r.get_path()
    with pytest.raises(engine_mod.PPPError):
        r.gather_waypoints(0, 0, 1, 0, [1], buf.ptr)         # a slice-range handle has no finished list


def test_gather_waypoints_multi_rank_pattern_through_a_recording_rccl(tmp_path):
    """The send/recv branch of ppp_gather_waypoints through the real entry point: the engine dlopens its RCCL, so a
    recording stand-in (PPP_RCCL_LIB, tests/helpers/rccl_standin.c) shows what a rank would enqueue -- blocks in rank
    order, the root's own block copied in place, a rank without rows skipped, the handle's stream and the caller's
    communicator passed through, the group closed when a transfer fails.  (The transfers themselves need several GPUs.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "gather_standin_drive.py"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().split("\n")[-1])
    W, base, stream = d["W"], d["recv_ptr"], d["stream"]
    assert d["own_block_in_place"]
    tail = "dtype=7 peer=%d comm=4660 stream=%d"
    assert d["root_log"] == ["group_start",
                             "recv buf=%d count=42 " % base + tail % (0, stream),
                             "recv buf=%d count=66 " % (base + 24 * (7 + W)) + tail % (3, stream),
                             "group_end"]
    assert len(d["send_log"]) == 3 and d["send_log"][0] == "group_start" and d["send_log"][2] == "group_end"
    assert d["send_log"][1].startswith("send buf=") and d["send_log"][1].endswith("count=%d " % (6 * W) + tail % (2, stream))
    assert d["fail_raised"] and "ncclResult 5" in d["fail_msg"]
    assert d["fail_log"][0] == "group_start" and d["fail_log"][-1] == "group_end" and len(d["fail_log"]) == 3   # no second recv after the failure


def test_gather_rehearsal_through_the_real_rccl():
    """Pre-flight of ppp_gather_waypoints' send / recv group against the REAL librccl on one GPU: a one-rank
    communicator handed to ppp_gather_waypoints with nranks == 1, the handle's list sent to itself and received inside one ncclGroupStart / ncclGroupEnd on
    the handle's stream; the received block equals the list byte for byte.  In a child process with a deadline: a collective
    that never completes must not take the suite with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from polishpathplanning_amd import engine, synth
from polishpathplanning_amd.hipbuf import DeviceBuffer
from polishpathplanning_amd.robot_path import RcclComm
pts, cfg = synth.make_config("small_40k")
e = engine.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.gen_path(); W = e.get_path()
own = e.waypoints()
comm = RcclComm(0, 1)
recv = DeviceBuffer(W * 24)
plain = DeviceBuffer(W * 24)
e.gather_waypoints(0, 0, 1, 0, [W], plain.ptr); e.sync()          # the lone rank's plain copy
for _ in range(3):                                                   # the group path, more than once on the same communicator
    e.gather_waypoints(comm.ptr, 0, 1, 0, [W], recv.ptr)
e.sync()
got = recv.to_host(W * 6).reshape(-1, 6)
print(json.dumps({"W": int(W), "equal": bool(got.tobytes() == own.tobytes()), "plain_equal": bool(plain.to_host(W * 6).reshape(-1, 6).tobytes() == own.tobytes())}))
comm.close()
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])
    assert d["W"] > 0 and d["equal"] and d["plain_equal"], d


def test_bench_one_rank_rehearsal_reports_what_the_collective_saw(tmp_path):
    """bench.py's N > 1 path on one GPU (PPP_BENCH_FORCE_DIST=1: a one-rank RCCL group): the JSON line carries the ranks the
    communicator saw, every rank's waypoint count and the gather's own time per step, next to the usual fields."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(PPP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "small_40k", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--rotate", "0", "--no-dynamic", "--profile-passes", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])
    m = d["multi_gpu"]
    assert m["n_ranks_seen"] == {"torch_distributed_world_size": 1, "rccl_allreduce_of_ones": 1}
    assert m["waypoints_per_rank"] == [d["config"]["waypoints_per_workpiece"]] and m["gather_ms_per_step_alone"] > 0
    assert d["assembled_path"]["rank0_block_equals_its_list"] and d["roofline"]["kernel"].startswith("k_")


@pytest.mark.parametrize("name", ["small_40k", "cfg3_250k_s128", "cfg2_1m_s256"])
def test_side_by_side_hint_changes_the_launch_not_the_list(engine_mod, name):
    """ppp_set_side_by_side(h, 3): the plan keeps the slice workgroups of small windows at 512 threads (room on a CU for the launches of
    the passes next door).  Knots, list and stage outputs are those of the default plan, byte for byte -- set before the cloud, after
    it, and taken back."""
    pts, cfg = synth.make_config(name)
    a = engine_mod.Engine(0, tool_radius=cfg["tool_radius"]); a.set_cloud(pts); a.run_async(); a.sync()
    b = engine_mod.Engine(0, tool_radius=cfg["tool_radius"]); b.set_side_by_side(3); b.set_cloud(pts); b.run_async(); b.sync()
    def same(x, y):
        assert x.num_slices() == y.num_slices() and x.waypoints().tobytes() == y.waypoints().tobytes()
        assert x.stage(engine_mod.STAGE_WP_NN).tobytes() == y.stage(engine_mod.STAGE_WP_NN).tobytes()
        for s in (0, x.num_slices() // 2, x.num_slices() - 1):
            assert all(np.array_equal(p, q) for p, q in zip(x.nodes(s), y.nodes(s)))
    same(a, b)
    a.set_side_by_side(2); a.run_async(); a.sync(); same(a, b)        # told later: planned again, same list
    b.set_side_by_side(1); b.run_async(); b.sync(); same(a, b)        # and taken back
    assert a.fast_path() and b.fast_path()


def test_planner_queue_lists_equal_fresh_handles(engine_mod):
    """ppp_queue_*: a stream of workpieces through three lanes (handles) taking turns -- same-size clouds (handed over without a wait for
    their bounds, their passes overlapping on the device), a cloud of another size in between, one without a single finite point.
    Every ticket's list is the list a fresh handle plans for that cloud, byte for byte; the empty cloud ends in its own error and
    the queue goes on; a ticket whose lane has been given a later workpiece is refused."""
    from polishpathplanning_amd.hipbuf import DeviceBuffer
    import ctypes as C
    base, cfg = synth.make_config("small_40k")
    clouds = [synth.make_config("small_40k", seed=300 + k)[0] for k in range(8)]
    clouds[3] = synth.make_config("tiny_5k")[0]                       # another size: planned from scratch in its lane
    clouds[5] = np.full_like(clouds[5], np.nan)                        # nothing to plan
    bufs = []
    for c in clouds:
        b = DeviceBuffer(c.nbytes); b.upload(np.ascontiguousarray(c, np.float32)); bufs.append(b)
    q0 = engine_mod.PlannerQueue(0, tool_radius=6.0)
    assert q0.lanes == 2                                                       # the default: a stream of new clouds
    q0.close()
    q = engine_mod.PlannerQueue(0, lanes=3, tool_radius=6.0)
    assert q.lanes == 3
    tickets = [q.submit(bufs[k].ptr, len(clouds[k])) for k in range(3)]       # three in flight
    results = {}
    for k in range(3, len(clouds) + 3):
        t = tickets[k - 3]
        try:
            W, dptr = q.wait(t)
            tmp = DeviceBuffer(max(W, 1) * 24); 
            from polishpathplanning_amd.hipbuf import _rt
            assert _rt().hipMemcpy(C.c_void_p(tmp.ptr), C.c_void_p(dptr), C.c_size_t(W * 24), 3) == 0
            results[t] = tmp.to_host(6 * W).reshape(-1, 6)
            tmp.free()
        except engine_mod.PPPError as ex:
            results[t] = ("error", ex.code)
        if k < len(clouds):
            tickets.append(q.submit(bufs[k].ptr, len(clouds[k])))
    assert tickets == list(range(len(clouds)))
    with pytest.raises(engine_mod.PPPError):
        q.wait(tickets[1])                                                     # its lane has planned two workpieces since
    for k, c in enumerate(clouds):
        f = engine_mod.Engine(0, tool_radius=6.0); f.set_plan_reuse(False)
        try:
            f.set_cloud(c); f.run_async(); f.sync()
            want = f.waypoints()
        except engine_mod.PPPError as ex:
            want = ("error", ex.code)
        f.close()
        got = results[tickets[k]]
        if isinstance(want, tuple):
            assert got == want, (k, got, want)
        else:
            assert not isinstance(got, tuple) and got.shape == want.shape and got.tobytes() == want.tobytes(), k
    q.close()
    for b in bufs:
        b.free()


def test_bench_line_of_a_one_gpu_run(tmp_path):
    """The default form of bench.py on one GPU (a small workload here): ONE JSON line with the contract's fields; consecutive steps take
    turns on three engine handles, every replica's list is the first handle's byte for byte, the one-handle loop is timed beside it, the
    oracle is the checker (path_l2_err) and the CPU baseline, and the roofline object names the dominant kernel with its duration alone
    on the device and under the loop's overlap."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "small_40k", "--steps", "12", "--warmup", "3", "--rotate", "2",
                        "--no-other-configs", "--profile-passes", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.strip()]
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["vs_baseline"] is None and d["unit"] == "waypoints/s" and d["higher_is_better"] is True
    assert d["config"]["handles_taking_turns"] == 3 and d["single_handle"]["replicas_equal"] is True
    assert abs(d["value"] - d["config"]["waypoints_per_workpiece"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["single_handle"]["ms_per_step"] > 0 and d["assembled_path"]["rank0_block_equals_its_list"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"].startswith("k_win_") and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["avg_launch_ms_steps_taking_turns"] > 0 and 0 < rf["pipeline_frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert d["path_l2_err"]["waypoints_equal"] and d["path_l2_err"]["max_m"] <= TOL_M
    assert d["dynamic"]["replicas_equal"] and d["dynamic"]["ms_per_step_three_handles"] > 0
    assert d["latency"]["cold_ms"] > 0


@pytest.mark.parametrize("name,walk,kw", [("small_40k", 1, {}), ("small_40k", 0, {}), ("small_40k", 2, {}), ("small_40k", 3, {}), ("small_40k", 4, {}),
                                          ("cfg3_250k_s128", 1, {}), ("small_40k", 1, dict(trim=5.0, drop_ends=0, smooth=0)),
                                          ("small_40k", 1, dict(path_resolution=3.0, rpy_resolution=0.0)), ("small_40k", 1, dict(change_range=0)),
                                          ("small_40k", 3, dict(pairing=1)), ("small_40k", 4, dict(pairing=1)), ("cfg1_50k_s32", 3, dict(pairing=1)),
                                          ("cfg3_250k_s128", 3, dict(pairing=1))])
def test_window_path_and_slab_path_agree(engine_mod, name, walk, kw):
    """The two launch sequences (ppp_set_fast_path) plan the same cloud: the same slices, knots, sampled waypoints and nearest
    cloud points bit for bit, the finished list within the float floor of the normals' summation order -- with the kd pairing
    and with v1's brute-force greedy pairing (pairing=1: Path_Generation.cpp:107-206, walks 3 and 4)."""
    pts, cfg = synth.make_config(name)
    if kw.get("change_range") == 0:
        pts = pts * np.float32(1000.0)
    unit = 1000.0 if kw.get("change_range") == 0 else 1.0
    a = engine_mod.Engine(0, tool_radius=cfg["tool_radius"], walk=walk, **kw); a.set_cloud(pts)
    b = engine_mod.Engine(0, tool_radius=cfg["tool_radius"], walk=walk, fast_path=False, **kw); b.set_cloud(pts)
    assert a.fast_path() and not b.fast_path()
    S = a.gen_path(); W = a.get_path()
    assert (S, W) == (b.gen_path(), b.get_path())
    assert np.array_equal(a.slice_positions(), b.slice_positions())
    for s in range(S):
        assert all(np.array_equal(x, y) for x, y in zip(a.nodes(s), b.nodes(s))), s
    assert np.array_equal(a.tail_index(), b.tail_index()) and np.array_equal(a.waypoint_counts(), b.waypoint_counts())
    assert np.array_equal(a.stage(engine_mod.STAGE_WP_XYZ), b.stage(engine_mod.STAGE_WP_XYZ))
    assert np.array_equal(a.stage(engine_mod.STAGE_WP_NN), b.stage(engine_mod.STAGE_WP_NN))
    na, nb = a.stage(engine_mod.STAGE_WP_NORMAL), b.stage(engine_mod.STAGE_WP_NORMAL)
    assert np.abs(na - nb).max() <= 2e-6
    assert np.abs(a.waypoints()[:, :3] - b.waypoints()[:, :3]).max() <= 1e-6 * unit
    assert a.fast_path()                                   # nothing was handed back
    mn, mx = a.minmax(); mn2, mx2 = b.minmax()
    assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
    # the same pass again as one graph replay, and the slab index on demand behind it: the finished list stays what it was
    want = a.waypoints().tobytes()
    a.run_async(); a.sync(); a.run_async(); a.sync()
    assert a.waypoints().tobytes() == want
    assert np.array_equal(a.slice_indices(S // 2), b.slice_indices(S // 2))
    assert a.waypoints().tobytes() == want and a.num_waypoints() == W


def test_window_path_hands_a_pass_back_when_a_search_leaves_its_window(engine_mod, oracle_mod):
    """A hole in the cloud wider than the windows' reach: the nearest point of a waypoint over the hole lies beyond its window,
    the pass is repeated on the slab index by itself, and the list is the slab path's (and the oracle's)."""
    pts, cfg = synth.make_config("small_40k")
    x, y = pts[:, 0] * 1000.0, pts[:, 1] * 1000.0
    probe = engine_mod.Engine(0, tool_radius=6.0); probe.set_cloud(pts)
    px = probe.slice_positions()
    cx = float(px[len(px) // 2])
    # a 18 mm x 18 mm hole centred on a plane: the waypoints sampled across it have their nearest cloud point ~9 mm away
    holed = np.ascontiguousarray(pts[~((np.abs(x - cx) < 9.0) & (np.abs(y - 10.0) < 9.0))])
    a = engine_mod.Engine(0, tool_radius=6.0); a.set_cloud(holed)
    b = engine_mod.Engine(0, tool_radius=6.0, fast_path=False); b.set_cloud(holed)
    assert a.fast_path()
    Sa, Wa = a.gen_path(), a.get_path()
    assert (Sa, Wa) == (b.gen_path(), b.get_path())
    assert a.waypoints().tobytes() == b.waypoints().tobytes()
    o = oracle_mod.Oracle(holed, tool_radius=6.0); o.gen_path(); o.get_path()
    assert np.linalg.norm(a.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1).max() <= TOL_M
    assert not a.fast_path()                               # this cloud and these parameters stay on the slab index ...
    a.set_cloud(pts)
    assert a.fast_path()                                   # ... a new cloud gets the window path again


def test_plan_reuse_for_a_stream_of_clouds_of_one_size(engine_mod, oracle_mod):
    """A handle that is fed one cloud after the other (same point count, same parameters) plans the first from a census of its
    windows and lets the later ones inherit those capacities (ppp_set_plan_reuse, default on): their lists equal the lists of
    fresh handles bit for bit; a later cloud whose windows do NOT fit the inherited capacities (the same number of points, piled
    into a few windows) is planned again from its own census by itself -- still on the window path, still the right list."""
    pts0, cfg = synth.make_config("small_40k")
    handle = engine_mod.Engine(0, tool_radius=6.0)
    for seed in (None, 77, 78, 79):
        pts = pts0 if seed is None else synth.make_config("small_40k", seed=seed)[0]
        handle.set_cloud(pts); S = handle.gen_path(); W = handle.get_path()
        fresh = engine_mod.Engine(0, tool_radius=6.0); fresh.set_plan_reuse(False)
        fresh.set_cloud(pts); assert (S, W) == (fresh.gen_path(), fresh.get_path())
        assert handle.fast_path() and fresh.fast_path()
        assert handle.waypoints().tobytes() == fresh.waypoints().tobytes()
        for s in (1, S // 2, S - 2):
            assert all(np.array_equal(x, y) for x, y in zip(handle.nodes(s), fresh.nodes(s)))
        fresh.close()
    # the same number of points, but 150 of them moved into the x range of one window (~560 points): it overflows the inherited plan
    piled = pts0.copy()
    px = handle.slice_positions()
    x = piled[:, 0] * 1000.0
    far = np.nonzero((x > px[len(px) * 3 // 4]) & (x < x.max() - 20.0) & (np.abs(x[:, None] - px[None, :]).min(axis=1) > 4.6))[0][:150]  # from between the windows
    rng = np.random.default_rng(5)
    donors = np.nonzero(np.abs(x - px[len(px) // 3]) < 3.5)[0]
    src = piled[rng.choice(donors, len(far))]
    piled[far, 0] = src[:, 0] + rng.uniform(-2e-4, 2e-4, len(far)).astype(np.float32)
    piled[far, 1] = rng.uniform(piled[:, 1].min(), piled[:, 1].max(), len(far)).astype(np.float32)
    piled[far, 2] = ((20.0 * np.sin(piled[far, 0].astype(np.float64) * 1000.0 / 600.0) * np.cos(piled[far, 1].astype(np.float64) * 1000.0 / 300.0) + 1500.0) / 1000.0).astype(np.float32)  # on the surface (synth: wavy, amp 20)
    handle.set_cloud(piled)
    fresh = engine_mod.Engine(0, tool_radius=6.0); fresh.set_plan_reuse(False); fresh.set_cloud(piled)
    try:
        ok_h = (handle.gen_path(), handle.get_path())
    except engine_mod.PPPError as ex_h:
        ok_h = str(ex_h)
    try:
        ok_f = (fresh.gen_path(), fresh.get_path())
    except engine_mod.PPPError as ex_f:
        ok_f = str(ex_f)
    assert type(ok_h) == type(ok_f)
    assert isinstance(ok_h, tuple) and ok_h == ok_f and handle.waypoints().tobytes() == fresh.waypoints().tobytes()
    assert handle.fast_path() and fresh.fast_path()        # planned again from its own census, still on the window path


def test_pass_enqueued_ahead_of_the_new_clouds_bounds(tmp_path):
    """A handle that holds a window plan of an earlier cloud of the same size and parameters need not wait for a new cloud's
    bounds: ppp_set_cloud_device_async enqueues the conversion pass and returns, run_async may follow at once on the earlier plan, and walk length,
    pad, bounds and capacities are checked on the device against the record the conversion pass leaves.  Whatever the new cloud
    looks like -- the same kind, shifted by two and a half slices, taller (more waypoints per slice than the plan has slots),
    wider (another slice count), with dropped points, all points dropped -- the list is the one a fresh, waiting handle makes,
    byte for byte; so are the answers of calls that need the host's view of the cloud before any pass ran."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from polishpathplanning_amd import engine, synth
        from polishpathplanning_amd.hipbuf import DeviceBuffer
        base, cfg = synth.make_config("small_40k")
        dbuf = DeviceBuffer(base.nbytes)
        def set_async(h, p):                  # the cloud in device memory, handed over without a wait (dbuf is rewritten only after a waiting call)
            dbuf.upload(np.ascontiguousarray(p, np.float32))
            h.set_cloud_device_async(dbuf.ptr, len(p), 12)
        def variant(kind):
            p = synth.make_config("small_40k", seed=100 + kind)[0].copy()
            if kind == 1: p[:, 0] += np.float32(0.031)                      # two and a half slices along x
            if kind == 2: p[:, 1] *= np.float32(1.25)                       # taller: more waypoints per slice
            if kind == 3: p[:, 0] *= np.float32(1.12)                       # wider: more slices
            if kind == 4: p[::53] = np.nan                                  # dropped points
            if kind == 5: p[:, 0] *= np.float32(0.8)                        # narrower: fewer slices, fuller windows
            if kind == 6: p[:, 2] += np.float32(0.2)                        # another height
            return p
        def fresh_result(p, **par):
            f = engine.Engine(0, tool_radius=6.0, **par); f.set_plan_reuse(False)
            f.set_cloud(p)
            try:
                f.run_async(); f.sync()
                r = (f.num_slices(), f.waypoints().tobytes(), f.slice_positions().tobytes(), [x.tobytes() for x in f.nodes(3)], f.fast_path())
            except engine.PPPError as ex:
                r = ("error", ex.code)
            f.close()
            return r
        h = engine.Engine(0, tool_radius=6.0)
        h.set_cloud(base); h.run_async(); h.sync()
        deferred = 0
        for rnd, kind in enumerate([0, 7, 1, 8, 2, 9, 3, 10, 4, 11, 5, 12, 6, 13, 0]):
            p = variant(kind)
            set_async(h, p)                     # no wait from the second same-size cloud on
            h.run_async()                       # ... and the pass right behind it
            try:
                h.sync()
                got = (h.num_slices(), h.waypoints().tobytes(), h.slice_positions().tobytes(), [x.tobytes() for x in h.nodes(3)], h.fast_path())
            except engine.PPPError as ex:
                got = ("error", ex.code)
            want = fresh_result(p)
            assert got == want, (rnd, kind, got[0], want[0])
        # the host's view before any pass: bounds, slice positions, nearest point, then the pass
        for kind in (14, 3, 15):
            p = variant(kind)
            f = engine.Engine(0, tool_radius=6.0); f.set_plan_reuse(False); f.set_cloud(p)
            set_async(h, p)
            assert all(np.array_equal(a, b) for a, b in zip(h.minmax(), f.minmax()))
            q = p[::1000] * 1000 + np.float32(0.01)
            assert np.array_equal(h.nearest(q), f.nearest(q))
            h.run_async(); f.run_async(); h.sync(); f.sync()
            assert h.waypoints().tobytes() == f.waypoints().tobytes()
            f.close()
        # two clouds set in a row, parameters changed under a cloud that was not waited for, preprocessing right after it
        set_async(h, variant(16)); h.sync(); set_async(h, variant(17)); h.run_async(); h.sync()
        assert (h.num_slices(), h.waypoints().tobytes()) == fresh_result(variant(17))[:2]
        set_async(h, variant(18)); h.set_params(tool_radius=7.0); h.run_async(); h.sync()
        f = engine.Engine(0, tool_radius=7.0); f.set_plan_reuse(False); f.set_cloud(variant(18)); f.run_async(); f.sync()
        assert h.waypoints().tobytes() == f.waypoints().tobytes(); f.close()
        h.set_params(tool_radius=6.0); h.set_cloud(variant(19)); h.run_async(); h.sync()
        set_async(h, variant(20)); n_left = h.remove_outlier(50, 1.0)[0]; h.run_async(); h.sync()
        f = engine.Engine(0, tool_radius=6.0); f.set_plan_reuse(False); f.set_cloud(variant(20)); assert f.remove_outlier(50, 1.0)[0] == n_left
        f.run_async(); f.sync()
        assert h.waypoints().tobytes() == f.waypoints().tobytes(); f.close()
        # GenPath / getPath as separate calls, and a replay on the inherited plan before anything was asked
        set_async(h, variant(21)); h.gen_path_async(); h.get_path_async(); h.run_async(); h.run_async(); h.sync()
        assert (h.num_slices(), h.waypoints().tobytes()) == fresh_result(variant(21))[:2]
        # a captured pass (two run_async on one plan) and then a cloud seen from the other side: the viewpoint travels by value in the
        # launches, so the capture must not be replayed for it
        h.set_cloud(variant(22)); h.run_async(); h.run_async(); h.sync()
        p = variant(23); dbuf.upload(np.ascontiguousarray(p, np.float32))
        h.set_cloud_device_async(dbuf.ptr, len(p), 12, viewpoint=[0.0, 0.0, 5000.0]); h.run_async(); h.sync()
        f = engine.Engine(0, tool_radius=6.0); f.set_plan_reuse(False); f.set_cloud(p, viewpoint=[0.0, 0.0, 5000.0]); f.run_async(); f.sync()
        g = engine.Engine(0, tool_radius=6.0); g.set_cloud(p); g.run_async(); g.sync()
        assert h.waypoints().tobytes() == f.waypoints().tobytes() and h.waypoints().tobytes() != g.waypoints().tobytes()
        f.close(); g.close()
        print("same lists")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPP_WIN_DEBUG="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "same lists" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # the taller and the wider clouds did not fit the plan they were enqueued on: handed back or repeated, by the engine itself
    assert "handed back" in r.stderr or "repeated" in r.stderr, r.stderr[-3000:]


def test_census_that_comes_with_a_new_cloud_equals_the_one_taken_at_plan_time(tmp_path):
    """A cloud that has just been set brings bounds, slice walk and window census along in the same stream
    (k_ingest_minmax's last workgroup + k_win_census_auto, results in pinned memory); a plan made later for the same cloud
    (new parameters) takes its census the earlier way.  Both must size the same plan and give the same list -- for all five
    walks, a cloud with dropped (NaN) points, and when a second cloud follows on the same handle."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from polishpathplanning_amd import engine, synth
        pts, cfg = synth.make_config("small_40k")
        pts2, _ = synth.make_config("small_40k", seed=77)
        pts2 = pts2.copy(); pts2[::97] = np.nan
        for walk in range(5):
            e = engine.Engine(0, tool_radius=6.0, walk=walk)
            e.set_plan_reuse(False)                                       # (every cloud takes its own census here: plan reuse has its own test)
            for cloud in (pts, pts2):
                e.set_cloud(cloud); e.gen_path(); e.get_path()          # census came with the cloud
                a = e.waypoints().copy(); assert e.fast_path()
                e.set_params(tool_radius=7.0); e.gen_path(); e.get_path()
                e.set_params(tool_radius=6.0); e.gen_path(); e.get_path()  # census at plan time
                b = e.waypoints().copy(); assert e.fast_path()
                assert a.shape == b.shape and np.array_equal(a, b), walk
        print("same lists")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPP_WIN_DEBUG="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "same lists" in r.stdout, r.stdout + r.stderr
    plans = [ln for ln in r.stderr.splitlines() if ln.startswith("[ppp] window plan")]
    assert len(plans) == 5 * 2 * 3
    for k in range(0, len(plans), 3):
        with_cloud, _, later = plans[k:k + 3]
        assert with_cloud.endswith("census came with the cloud") and later.endswith("census at plan time")
        assert with_cloud.rsplit(" census ", 1)[0] == later.rsplit(" census ", 1)[0]   # capacities, buckets, threads: the same plan


def test_window_path_applies_only_where_the_windows_do_not_overlap(engine_mod, oracle_mod):
    """Tool steps below about 2 x pad + 2 mm (here radius 4 -> step 8) make the slices' windows overlap: the plan stays on the
    slab index, with the same parity; so does the dynamic adjustment.  (Brute pairing runs on the window path since round 4.)"""
    pts, cfg = synth.make_config("small_40k")
    for kw in (dict(tool_radius=4.0), dict(tool_radius=6.0, dynamic_adjustment=1), dict(tool_radius=6.0, pairing=1, walk=3, dynamic_adjustment=1)):
        e = engine_mod.Engine(0, **kw); e.set_cloud(pts)
        assert not e.fast_path()
    e = engine_mod.Engine(0, tool_radius=6.0, pairing=1, walk=3); e.set_cloud(pts)
    assert e.fast_path()
    e, o = run_pair(engine_mod, oracle_mod, pts, tool_radius=4.0)
    assert_full_parity(engine_mod, e, o, every_slice=False)
    assert engine_mod.Engine(0, tool_radius=5.0).fast_path() is False     # no cloud yet: nothing planned
