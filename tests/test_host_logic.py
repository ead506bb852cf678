"""Host-side logic that needs no GPU: synthetic configs, path file format, and the N > 1
robot_path gather (world_size 2, gloo)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_seeded_and_in_metres():
    from polishpathplanning_amd import synth
    a, cfg = synth.make_config("tiny_5k")
    b, _ = synth.make_config("tiny_5k")
    assert a.dtype == np.float32 and a.shape == (5000, 3) and np.array_equal(a, b)
    assert 1.4 < a[:, 2].mean() < 1.6  # 1.5 m in front of the sensor
    c, _ = synth.make_config("tiny_5k", seed=99)
    assert not np.array_equal(a, c)


@pytest.mark.parametrize("name", ["cfg1_50k_s32", "cfg3_250k_s128"])
def test_configs_hit_their_slice_count(oracle_mod, name):
    from polishpathplanning_amd import synth
    pts, cfg = synth.make_config(name)
    o = oracle_mod.Oracle(pts, tool_radius=cfg["tool_radius"], walk=1)
    assert len(o.slice_positions()) == cfg["slices"]


def test_path_file_format(tmp_path):
    from polishpathplanning_amd.robot_path import write_path_file
    wp = np.array([[0.5, -0.25, 1.0, 3.14159274, -1e-5, 123456.789], [1e-7, 2, 3, 4, 5, 6]], np.float32)
    p = tmp_path / "WayPoints.txt"
    write_path_file(str(p), wp)
    lines = p.read_text().split("\n")
    assert lines[0] == "0.5 -0.25 1 3.14159 -1e-05 123457 "   # ostream default precision, trailing blank
    assert lines[1] == "1e-07 2 3 4 5 6 " and lines[2] == ""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polishpathplanning_amd.robot_path import gather_robot_path, concat_robot_path
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = 5 + 3 * rank  # ragged blocks
    local = torch.arange(w * 6, dtype=torch.float32).reshape(w, 6) + 1000 * rank
    blocks = gather_robot_path(local, dist)
    from polishpathplanning_amd.robot_path import exchange_counts
    counts = exchange_counts(w, dist, local.device)
    again = gather_robot_path(local, dist, counts=counts)       # cached counts: no count exchange
    from polishpathplanning_amd.robot_path import RobotPathGatherer
    g = RobotPathGatherer(w, dist, local.device)                # preallocated form: the planner writes into g.send
    assert g.counts == counts and g.send.shape == (max(counts), 6)
    third = None
    for _ in range(2):                                          # buffers are reused from step to step
        g.send[:w] = local
        third = g.gather()
    if rank == 0:
        full = concat_robot_path(blocks)
        assert torch.equal(full, concat_robot_path(again))
        assert torch.equal(full, concat_robot_path(third))
        q.put(([b.shape[0] for b in blocks], full.numpy()))
    else:
        assert blocks is None and again is None and third is None
    dist.barrier()
    dist.destroy_process_group()


def test_robot_path_gather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert counts == [5, 8]
    want = np.concatenate([np.arange(30, dtype=np.float32).reshape(5, 6),
                           np.arange(48, dtype=np.float32).reshape(8, 6) + 1000])
    assert np.array_equal(full, want)


def _slice_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polishpathplanning_amd.robot_path import gather_slice_blocks, slice_ranges
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, first_kept, nkept = 11, 1, 9              # drop_ends: slices 1..9 are kept
    per_slice = np.array([3 + (k % 4) for k in range(nkept)], np.int32)
    b, e = slice_ranges(S, world)[rank]
    counts = np.array([per_slice[k] if b <= k + first_kept < e else 0 for k in range(nkept)], np.int32)
    rows = []
    for k in range(nkept):                       # waypoint value = 100 * kept slice + position
        rows += [[100.0 * k + t] * 6 for t in range(counts[k])]
    local = torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)
    pre, cnt = gather_slice_blocks(local, counts, dist)
    if rank == 0:
        q.put((pre.numpy(), cnt))
    else:
        assert pre is None and cnt is None
    dist.barrier()
    dist.destroy_process_group()


def test_slice_range_gather_world2_gloo():
    """SURVEY.md 8e case (ii): ranks hold disjoint slice ranges; rank 0 gets the list in slice order plus
    the global per-slice counts it needs for TailIndex."""
    import torch.multiprocessing as mp
    from polishpathplanning_amd.robot_path import slice_ranges
    assert slice_ranges(11, 2) == [(0, 5), (5, 11)] and slice_ranges(1024, 8)[7] == (896, 1024)
    assert slice_ranges(3, 8)[0] == (0, 0)       # more GPUs than slices: empty ranges are legal
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pre, cnt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    per_slice = np.array([3 + (k % 4) for k in range(9)], np.int32)
    assert np.array_equal(cnt, per_slice)
    want = np.concatenate([[100.0 * k + t for t in range(per_slice[k])] for k in range(9)]).astype(np.float32)
    assert np.array_equal(pre[:, 0], want) and pre.shape == (per_slice.sum(), 6)


def _pipeline_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polishpathplanning_amd.robot_path import RobotPathGatherer, run_pipelined_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = 4 + 5 * rank                                     # ragged: rank 0 has 4 waypoints, rank 1 has 9
    gatherers = [RobotPathGatherer(w, dist, torch.device("cpu")) for _ in range(2)]
    seen = []

    def plan(k):                                         # "planner": step k's list = 100 k + 1000 rank + row
        g = gatherers[k % 2]
        g.send[:w] = (torch.arange(w, dtype=torch.float32)[:, None] + 100.0 * k + 1000.0 * rank).expand(w, 6)

    def on_blocks(k, blocks):
        if rank == 0:
            seen.append((k, [b.clone() for b in blocks]))
        else:
            assert blocks is None

    from polishpathplanning_amd.robot_path import NoOrder, run_streamed_steps

    class Recorder(NoOrder):                             # the order in which bench.py's event hooks are called
        def __init__(self): self.calls = []
        def before_plan(self, b): self.calls.append(("bp", b))
        def after_plan(self, b): self.calls.append(("ap", b))
        def before_gather(self, b): self.calls.append(("bg", b))
        def after_gather(self, b): self.calls.append(("ag", b))

    from polishpathplanning_amd.robot_path import run_chained_steps
    for count, streamed in [(0, False), (1, False), (2, False), (5, False), (0, True), (1, True), (2, True), (5, True),
                            (0, "chained"), (1, "chained"), (2, "chained"), (3, "chained"), (6, "chained")]:
        del seen[:]
        if streamed == "chained":          # bench.py's default loop: asynchronous collectives, waited for two steps later
            last = run_chained_steps(count, plan, gatherers, None, on_blocks)
        elif streamed:
            rec = Recorder()
            last = run_streamed_steps(count, plan, gatherers, rec, on_blocks)
            # every buffer: planned before it is gathered, gathered before it is planned again; the gather of step k-1
            # is enqueued behind the planning of step k
            want_calls = []
            for k in range(count):
                want_calls += [("bp", k % 2), ("ap", k % 2)]
                if k > 0:
                    want_calls += [("bg", (k - 1) % 2), ("ag", (k - 1) % 2)]
            if count:
                want_calls += [("bg", (count - 1) % 2), ("ag", (count - 1) % 2)]
            assert rec.calls == want_calls
        else:
            last = run_pipelined_steps(count, plan, lambda: None, gatherers, lambda: None, on_blocks)
        if rank == 0:
            assert [k for k, _ in seen] == list(range(count))
            for k, blocks in seen:
                assert [b.shape[0] for b in blocks] == [4, 9]
                for r, b in enumerate(blocks):
                    want = (torch.arange(b.shape[0], dtype=torch.float32)[:, None] + 100.0 * k + 1000.0 * r).expand(-1, 6)
                    assert torch.equal(b, want), (count, k, r)
            assert (last is None) == (count == 0)
        else:
            assert last is None
    # the same chain with THREE buffer pairs taken in turn (bench.py's steps on three engine handles, each with its own pair): every
    # step's blocks arrive once, in step order, from both ranks
    g3 = [RobotPathGatherer(w, dist, torch.device("cpu")) for _ in range(3)]

    def plan3(k):
        g3[k % 3].send[:w] = (torch.arange(w, dtype=torch.float32)[:, None] + 100.0 * k + 1000.0 * rank).expand(w, 6)
    for count in (0, 1, 2, 3, 4, 7):
        del seen[:]
        last = run_chained_steps(count, plan3, g3, [None, None, None], on_blocks)
        if rank == 0:
            assert [k for k, _ in seen] == list(range(count))
            for k, blocks in seen:
                for r, b in enumerate(blocks):
                    want = (torch.arange(b.shape[0], dtype=torch.float32)[:, None] + 100.0 * k + 1000.0 * r).expand(-1, 6)
                    assert b.shape[0] == [4, 9][r] and torch.equal(b, want), (count, k, r)
            assert (last is None) == (count == 0)
        else:
            assert last is None
    if rank == 0:
        q.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_steps_world2_gloo():
    """bench.py's N > 1 loops (event-ordered run_streamed_steps, and the earlier run_pipelined_steps with host waits): the
    gather of step k-1 overlaps the planning of step k over two buffer pairs; every step's blocks arrive intact and in
    order, for 0, 1, 2 and 5 steps; the ordering hooks are called in the order the buffers need."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) == "ok"
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def test_gather_without_process_group():
    import torch
    from polishpathplanning_amd.robot_path import gather_robot_path
    t = torch.ones((3, 6))
    assert gather_robot_path(t)[0] is t


# ---------------- file formats of the reference (host side of the C ABI) ----------------
@pytest.mark.parametrize("binary", [True, False])
def test_pcd_roundtrip(engine_mod, tmp_path, binary):
    from polishpathplanning_amd import synth
    pts, _ = synth.make_config("tiny_5k")
    pts = pts.copy()
    pts[3] = [np.nan, np.nan, np.nan]
    p = str(tmp_path / "c.pcd")
    engine_mod.save_pcd(p, pts, viewpoint=[0.1, 0.2, 0.3, 1, 0, 0, 0], binary=binary)
    back, vp = engine_mod.load_pcd(p)
    assert back.shape == pts.shape and np.array_equal(back[~np.isnan(back)], pts[~np.isnan(pts)])
    assert np.isnan(back[3]).all() and np.allclose(vp, [0.1, 0.2, 0.3, 1, 0, 0, 0])


def _lzf_literal_only(raw):
    """A second, independent LZF encoder (literal runs only) for the decoder test."""
    out = bytearray()
    for i in range(0, len(raw), 32):
        run = raw[i:i + 32]
        out.append(len(run) - 1)
        out += run
    return bytes(out)


def test_pcd_binary_compressed(engine_mod, tmp_path):
    """PCD DATA binary_compressed (what pcl::io::savePCDFileBinaryCompressed writes): LZF stream of the field-major block."""
    import struct
    rng = np.random.default_rng(7)
    pts = rng.normal(size=(5000, 3)).astype(np.float32)
    pts[::7] = pts[0]                                   # repeats: the encoder emits back references
    pts[:, 2] = 0.25                                    # a flat plate: the z block is one long run
    p = str(tmp_path / "c.pcd")
    engine_mod.save_pcd(p, pts, viewpoint=[1, 2, 3, 1, 0, 0, 0], binary="compressed")
    raw = open(p, "rb").read()
    assert b"DATA binary_compressed" in raw and len(raw) < 60000 * 0.75   # it did compress
    got, vp = engine_mod.load_pcd(p)
    assert np.array_equal(got, pts) and list(vp[:3]) == [1, 2, 3]
    # a stream from an independent encoder, with an rgb field between the coordinates (field-major layout) and double z
    n = 257
    x = rng.normal(size=n).astype(np.float32); y = rng.normal(size=n).astype(np.float32); z = rng.normal(size=n)
    rgb = rng.integers(0, 2 ** 24, size=n, dtype=np.uint32)
    block = x.tobytes() + rgb.tobytes() + y.tobytes() + z.astype(np.float64).tobytes()
    comp = _lzf_literal_only(block)
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS x rgb y z\nSIZE 4 4 4 8\nTYPE F U F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n" % (n, n)).encode()
    q = str(tmp_path / "d.pcd")
    open(q, "wb").write(hdr + struct.pack("<II", len(comp), len(block)) + comp)
    got, _ = engine_mod.load_pcd(q)
    assert np.array_equal(got, np.stack([x, y, z.astype(np.float32)], axis=1))
    # a back reference that points before the start of the output is rejected, not followed
    bad = str(tmp_path / "bad.pcd")
    open(bad, "wb").write(hdr + struct.pack("<II", 4, len(block)) + b"\xe0\x05\x00\x00")   # reference before the start
    with pytest.raises(engine_mod.PPPError):
        engine_mod.load_pcd(bad)


def test_pcd_with_extra_fields_and_double_xyz(engine_mod, tmp_path):
    # FIELDS in any order / F8 coordinates / rgb packed as U4, as PCL writes them
    n = 5
    rec = np.zeros(n, dtype=[("rgb", "<u4"), ("x", "<f8"), ("normal_x", "<f4"), ("y", "<f8"), ("z", "<f8")])
    rec["x"] = np.arange(n) * 0.001; rec["y"] = 0.5; rec["z"] = 1.5 + np.arange(n); rec["rgb"] = 0xffffff
    p = tmp_path / "d.pcd"
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS rgb x normal_x y z\nSIZE 4 8 4 8 8\nTYPE U F F F F\nCOUNT 1 1 1 1 1\n"
           "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    p.write_bytes(hdr.encode() + rec.tobytes())
    xyz, vp = engine_mod.load_pcd(str(p))
    assert np.allclose(xyz[:, 0], rec["x"]) and np.allclose(xyz[:, 1], 0.5) and np.allclose(xyz[:, 2], rec["z"])


def test_big_ascii_pcd_is_cut_into_pieces_at_line_ends(engine_mod, tmp_path):
    """Ascii files of 4 MB and more are parsed by several threads, the text cut at line ends: rows counted per piece first, then
    parsed to their places.  Blank and blank-only lines are no rows wherever they fall, rows beyond POINTS are ignored, a file
    with fewer rows than POINTS is refused -- as in the single walk small files take."""
    rng = np.random.default_rng(8)
    n = 260_000
    pts = (rng.standard_normal((n, 3)) * [1.5, 0.1, 0.02]).astype(np.float32)
    rows = ["%.9g %.9g %.9g 4.2e+06" % tuple(r) for r in pts]
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
           "POINTS %d\nDATA ascii\n")
    body = []
    blanks = set(rng.integers(0, n, 3000).tolist())
    for i, r in enumerate(rows):
        if i in blanks:
            body.append("" if i % 2 else "  \t ")
        body.append(r)
    p = tmp_path / "big.pcd"
    p.write_text(hdr % (n, n) + "\n".join(body) + "\n")
    assert p.stat().st_size > (8 << 20)
    xyz, _ = engine_mod.load_pcd(str(p))
    assert np.array_equal(xyz.view(np.uint32), pts.view(np.uint32))
    p.write_text(hdr % (n - 1000, n - 1000) + "\n".join(body) + "\n")      # more rows than POINTS
    xyz, _ = engine_mod.load_pcd(str(p))
    assert np.array_equal(xyz.view(np.uint32), pts[: n - 1000].view(np.uint32))
    p.write_text(hdr % (n + 1, n + 1) + "\n".join(body) + "\n")            # one row short
    with pytest.raises(engine_mod.PPPError):
        engine_mod.load_pcd(str(p))


def test_pcd_probe_reports_the_record_layout(engine_mod, tmp_path):
    """ppp_pcd_probe: the header alone -- what ppp_set_cloud_pcd decides on (records streamed straight to HBM only for
    `DATA binary` with x, y, z as consecutive float32 fields)."""
    n = 7
    rec = np.zeros(n, dtype=[("rgb", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("curvature", "<f4"), ("label", "<u2"), ("pad", "<u2")])
    p = tmp_path / "a.pcd"
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS rgb x y z curvature label pad\nSIZE 4 4 4 4 4 2 2\nTYPE U F F F F U U\nCOUNT 1 1 1 1 1 1 1\n"
           "WIDTH %d\nHEIGHT 1\nVIEWPOINT 1 2 3 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    p.write_bytes(hdr.encode() + rec.tobytes())
    lay = engine_mod.pcd_probe(str(p))
    assert (lay.data_kind, lay.points, lay.record_bytes, lay.x_offset, lay.y_offset, lay.z_offset, lay.xyz_float32) == (1, n, 24, 4, 8, 12, 1)
    assert lay.data_offset == len(hdr) and list(lay.viewpoint) == [1, 2, 3, 1, 0, 0, 0]
    rec8 = np.zeros(n, dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8")])
    hdr8 = "VERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA binary\n" % (n, n)
    p.write_bytes(hdr8.encode() + rec8.tobytes())
    lay = engine_mod.pcd_probe(str(p))
    assert (lay.data_kind, lay.record_bytes, lay.xyz_float32, lay.z_offset) == (1, 24, 0, 16)
    pts = np.arange(3 * n, dtype=np.float32).reshape(n, 3)
    for mode, kind in ((False, 0), (True, 1), ("compressed", 2)):
        engine_mod.save_pcd(str(p), pts, binary=mode)
        lay = engine_mod.pcd_probe(str(p))
        assert (lay.data_kind, lay.points, lay.record_bytes, lay.x_offset, lay.xyz_float32) == (kind, n, 12, 0, 1)
    p.write_bytes(hdr8.encode() + rec8.tobytes()[:-1])          # a record short: refused by the header check already
    with pytest.raises(engine_mod.PPPError):
        engine_mod.pcd_probe(str(p))
    with pytest.raises(engine_mod.PPPError):
        engine_mod.pcd_probe(str(tmp_path / "missing.pcd"))


def test_pcd_errors(engine_mod, tmp_path):
    with pytest.raises(engine_mod.PPPError) as ei:
        engine_mod.load_pcd(str(tmp_path / "missing.pcd"))
    assert ei.value.code == engine_mod.ERR_IO
    p = tmp_path / "z.pcd"
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n")
    with pytest.raises(engine_mod.PPPError) as ei:       # truncated: no size words, no stream
        engine_mod.load_pcd(str(p))
    assert ei.value.code == engine_mod.ERR_IO
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA zstd\n")
    with pytest.raises(engine_mod.PPPError) as ei:
        engine_mod.load_pcd(str(p))
    assert ei.value.code == engine_mod.ERR_UNSUPPORTED


def test_hostile_pcd_header_is_an_error_not_a_crash(engine_mod, tmp_path):
    """A header that promises more points than memory holds must come back as an error through the C ABI."""
    p = tmp_path / "huge.pcd"
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 900000000000\nHEIGHT 1\nPOINTS 900000000000\nDATA binary\n")
    with pytest.raises(engine_mod.PPPError) as ei:
        engine_mod.load_pcd(str(p))
    assert ei.value.code == engine_mod.ERR_IO
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 1\nPOINTS 5\nDATA binary_compressed\n")
    with open(p, "ab") as f:
        f.write(b"\xff\xff\xff\x7f\x3c\x00\x00\x00" + b"\x00" * 16)      # claims a 2 GiB stream
    with pytest.raises(engine_mod.PPPError):
        engine_mod.load_pcd(str(p))
    # SIZE / COUNT that would put x, y, z outside the record
    for hdr in ("FIELDS a x y z\nSIZE -8 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n", "FIELDS a x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT -3 1 1 1\n",
                "FIELDS x y z\nSIZE 0 4 4\nTYPE F F F\nCOUNT 1 1 1\n", "FIELDS a x y z\nSIZE 8 4 4 4\nTYPE F F F F\nCOUNT 2147483647 1 1 1\n"):
        p.write_bytes(("VERSION 0.7\n" + hdr + "WIDTH 5\nHEIGHT 1\nPOINTS 5\nDATA binary\n").encode() + b"\0" * 200)
        with pytest.raises(engine_mod.PPPError) as ei:
            engine_mod.load_pcd(str(p))
        assert ei.value.code == engine_mod.ERR_IO


def test_read_config_follows_the_reference_parser(engine_mod, tmp_path):
    ref = "/root/reference/config.txt"
    if os.path.exists(ref):  # only in the build container; the same text is restated below
        rc, c = engine_mod.read_config(ref)
        assert rc == 0 and c.params.tool_radius == 12 and c.path_file == b"WayPoints_test2.txt" and c.dynamic_adjustment == 1
    p = tmp_path / "config.txt"
    p.write_text("# comment\nTool_Radius = 6\npathFile = out dir/Way Points.txt\nEnd effector length = 0.25\n"
                 "ChangeRange = True\nDynamic_adjustment = false\nPathResolution=3.5\nbogus line\nRPYresolution = 9\nunknown = 1\n")
    rc, c = engine_mod.read_config(str(p))
    assert rc == 0
    assert c.params.tool_radius == 6 and c.params.path_resolution == 3.5 and c.params.rpy_resolution == 9
    assert abs(c.params.ee_length - 0.25) < 1e-7
    assert c.path_file == b"outdir/WayPoints.txt"   # every blank is stripped (path_slicing_alg.cpp:42)
    assert c.params.change_range == 0                # booleans are exactly "true" (:60)
    assert c.dynamic_adjustment == 0 and c.depth == 0.01  # absent keys keep config.txt's values
    rc, c = engine_mod.read_config(str(tmp_path / "nope.txt"))
    assert rc == engine_mod.ERR_IO and c.params.tool_radius == 12


def test_write_path_file_c_abi_matches_python_writer(engine_mod, tmp_path):
    from polishpathplanning_amd.robot_path import write_path_file
    wp = np.random.default_rng(0).normal(size=(50, 6)).astype(np.float32) * [1, 1, 1, 3, 3, 3]
    wp[0] = [0.5, -0.25, 1.0, 3.14159274, -1e-5, 123456.789]
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    engine_mod.write_path_file(a, wp)
    write_path_file(b, wp)
    assert open(a).read() == open(b).read()
    assert open(a).readline() == "0.5 -0.25 1 3.14159 -1e-05 123457 \n"


def test_path_file_bytes_are_printf_g_for_every_kind_of_float(engine_mod, tmp_path):
    """pathFile is `ofstream << float` (path_translation_alg.cpp:216-228), i.e. printf("%g") of the promoted value.  The writer
    formats with a routine of its own (ppp_io.cpp: format_g6): random bit patterns of every exponent, values that scale to exact
    and near ties at the sixth digit, powers of ten, denormals, infinities -- against Python's '%g' of the same double."""
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2**32, 240000, dtype=np.uint64).astype(np.uint32)
    v = bits.view(np.float32).copy()
    v[np.isnan(v)] = np.float32(np.nan)                       # (a NaN's sign shows as "-nan" in C and not in Python: positive only)
    ties = (rng.integers(1, 2000000, 60000) * 0.5).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 7, 60000).astype(np.float32)
    near = np.nextafter(ties, np.float32(np.inf) * np.where(rng.integers(0, 2, ties.size) == 1, 1, -1).astype(np.float32))
    special = np.float32([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, np.inf, -np.inf, 1e5, 1e6, 999999.5, 9999995, 99999.95,
                          0.0001, 0.00001, 0.000099999997, 123456.5, 1234565, 0.5, 1e21, 1e22, 1e-16, 9.9999994e-17, 1e15, 999999.44])
    p10 = (np.float64(10.0) ** np.arange(-45, 39)).astype(np.float32)
    allv = np.concatenate([v, ties, near, special, p10, np.nextafter(p10, np.float32(0)), np.nextafter(p10, np.float32(np.inf))])
    allv = np.concatenate([allv, np.zeros((-len(allv)) % 6, np.float32)]).reshape(-1, 6)
    out = str(tmp_path / "wp.txt")
    engine_mod.write_path_file(out, allv)
    want = "".join("".join("%g " % float(x) for x in row) + "\n" for row in allv)
    got = open(out).read()
    if got != want:
        g, w = got.split(), want.split()
        bad = [(a, b, float(x)) for a, b, x in zip(g, w, allv.ravel()) if a != b][:5]
        raise AssertionError("pathFile differs from printf %%g: %r" % (bad,))


def test_ascii_pcd_numbers_are_the_correctly_rounded_floats(engine_mod, tmp_path):
    """An ascii PCD goes through PCL's `istringstream >> float`, i.e. strtof: ONE rounding from the decimal text.  The loader's
    own parser (ppp_io.cpp: parse_float_token) against the C library's strtof, on nine-digit prints, long prints, decimal
    midpoints of neighbouring floats (exact ties), their neighbours, and the spellings strtof takes and a digit loop does not."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.strtof.restype = ctypes.c_float
    libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    rng = np.random.default_rng(6)
    f = (rng.standard_normal(20000) * 10.0 ** rng.integers(-5, 6, 20000)).astype(np.float32)
    toks = ["%.9g" % x for x in f[:6000]] + ["%.6e" % x for x in f[6000:9000]] + ["%f" % x for x in f[9000:12000]]
    from decimal import Decimal
    g = np.nextafter(f[12000:16000], np.float32(np.inf))
    for a, b in zip(f[12000:16000], g):                          # the midpoint of two neighbouring floats, all its digits: a tie
        mid = (Decimal(float(a)) + Decimal(float(b))) / 2
        toks.append(format(mid, "f") if abs(mid) > Decimal("1e-4") else "%.17g" % float(mid))
    for a, b in zip(f[16000:20000], np.nextafter(f[16000:20000], np.float32(np.inf))):   # ... and a double next to it
        mid = (float(a) + float(b)) / 2
        toks.append("%.17g" % np.nextafter(mid, np.inf if rng.integers(0, 2) else -np.inf))
    toks += ["nan", "NaN", "-inf", "inf", "0x1.8p1", "1e", ".5", "5.", "-0", "+2.5", "1e-50", "3.4028236e38", "1e39", "1.17549435e-38",
             "1e-45", "7e-46", "00012.500", "1E+2", "1e+0007", "0.000000000000000000001", "123456789012345678901234567890", "abc"]
    toks += ["0"] * ((-len(toks)) % 3)
    n = len(toks) // 3
    p = tmp_path / "t.pcd"
    p.write_text("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                 "POINTS %d\nDATA ascii\n" % (n, n) + "".join("%s\t%s  %s \r\n" % tuple(toks[3 * i:3 * i + 3]) for i in range(n)))
    xyz, _ = engine_mod.load_pcd(str(p))
    want = np.array([libc.strtof(t.encode(), None) for t in toks], np.float32)
    got = xyz.ravel()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), [(toks[i], float(want[i]), float(got[i])) for i in np.nonzero(~same)[0][:5]]


def test_pcd_and_config_parsers_under_address_sanitizer(tmp_path):
    """The host I/O of the C ABI (ppp_io.cpp: PCD ascii / binary / binary_compressed, LZF, config.txt, pathFile) compiled alone
    with -fsanitize=address,undefined and fed 4500 mutated / truncated PCD files and 800 mutated config files: every file is
    loaded or refused, nothing reads or writes out of bounds, no header makes it allocate beyond what the file can hold."""
    import subprocess
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    so = str(tmp_path / "libppp_io_asan.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-shared", "-o", so,
                           os.path.join(ROOT, "polishpathplanning_amd", "csrc", "ppp_io.cpp"), os.path.join(ROOT, "tests", "helpers", "io_stub.cpp")])
    work = tmp_path / "files"
    work.mkdir()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "io_sanitizer_drive.py"), so, str(work)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the form the driver uses for N = 1) must start the two ranks itself,
    before anything touches the GPU.  On this CPU-only box each rank then stops at the no-GPU check -- which is the proof
    that both were started with their RANK / WORLD_SIZE set and that the parent hands their failure on."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PPP_BENCH_ECHO_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    assert "rank 0 of 2" in out and "rank 1 of 2" in out
    assert "cfg4_2m_s256" in out          # the N > 1 default workload is BASELINE configs[3], one 2 M-point workpiece per GPU


@pytest.mark.parametrize("walk", [0, 1, 2, 3, 4])
def test_range_interval_is_the_walk_of_the_oracle(engine_mod, oracle_mod, walk):
    """ppp_range_interval (host arithmetic, no device): the x interval a slice-range handle indexes = the PassThrough bands of
    its first and last slice (rangedX_index(int): [int(Px) - 2, int(Px) + 2]) widened by range_margin, for the walk the
    oracle runs on a cloud with the same bounds."""
    import ctypes as C
    from polishpathplanning_amd import synth
    pts = synth.make_plate(300, 8, seed=5, x0_mm=-41.3)
    o = oracle_mod.Oracle(pts, tool_radius=7.5, walk=walk)
    mn, mx = o.minmax()
    px = o.slice_positions()
    L = engine_mod.lib()
    for b, e, margin in [(0, 0, 24.0), (3, 9, 24.0), (len(px) - 4, 0, 5.0), (2, 3, 30.0), (7, 7, 24.0)]:
        p = engine_mod.default_params(tool_radius=7.5, walk=walk, slice_begin=b, slice_end=e, range_margin=margin)
        lo, hi, S = C.c_float(), C.c_float(), C.c_int()
        assert L.ppp_range_interval(C.byref(p), float(mn[0]), float(mx[0]), C.byref(lo), C.byref(hi), C.byref(S)) == 0
        assert S.value == len(px)
        se = len(px) if e <= 0 else e
        if b == 0 and se == len(px):
            assert lo.value == -np.inf and hi.value == np.inf          # the whole walk: every point
        elif b >= se:
            assert lo.value > hi.value                                  # an empty range: no point
        else:
            assert lo.value == np.float32(int(px[b]) - 2) - np.float32(margin)
            assert hi.value == np.float32(int(px[se - 1]) + 2) + np.float32(margin)


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree exists in the build container only (never copied, never shipped to the GPU box)")
@pytest.mark.parametrize("driver,defs", [("main", []), ("connect", []), ("connect1", ["-DPPP_SDIR"]), ("contour", [])])
def test_reference_drivers_compile_unchanged(engine_mod, tmp_path, driver, defs):
    """the reference's own src/{main,connect,connect1,contour}.cpp compile UNCHANGED against
    include/ (which supplies the cout / endl names VTK leaks upstream) and link against libppp_hip.so.  The only thing
    this image lacks for them is PCL's console helper: tests/helpers/pcl_console_standin stands in for
    <pcl/console/parse.h> / <pcl/console/print.h>.  The sources are compiled where they lie; nothing is copied."""
    import subprocess
    src = os.path.join(REF_SRC, driver + ".cpp")
    exe = str(tmp_path / driver)
    libdir = os.path.dirname(engine_mod.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", *defs, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "helpers", "pcl_console_standin"),
           "-o", exe, src, "-L", libdir, "-lppp_hip", "-lX11", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # the drivers' own argument check (src/connect.cpp:11-21): no .pcd argument -> the usage line and -1, before any planner exists
    r = subprocess.run([exe, "not_a_cloud.txt"], capture_output=True, text=True)
    assert r.returncode == 255 and "./slicing_method cad_name.pcd" in r.stdout


def _gather_drive(workdir, *args):
    import subprocess
    exe = os.path.join(workdir, "gather_drive")
    if not os.path.exists(exe):
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "polishpathplanning_amd", "csrc"), "-o", exe,
                            os.path.join(ROOT, "tests", "helpers", "gather_exchange_drive.cpp")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().split("\n")


def test_gather_exchange_call_pattern(tmp_path):
    """The send/recv branch of ppp_gather_waypoints (its host logic, ppp_gather.h) under recording stand-ins for RCCL:
    block offsets in rank order, element counts, a root that is not rank 0, ranks without rows, and ncclGroupEnd
    reached on every path once ncclGroupStart has succeeded (the open-group early return)."""
    d = str(tmp_path)
    counts = [5, 0, 7, 3]
    # root = 2 of 4: its own block is copied (outside the group) to row offset 5, ranks 0 and 3 are received at 0 and 12, rank 1 has no rows
    out = _gather_drive(d, 2, 4, 2, -1, *counts)
    assert out[0].startswith("copy off=%d bytes=%d src_off=0" % (6 * 5, 24 * 7))
    assert out[1] == "group_start" and out[-2] == "group_end" and out[-1] == "rc=0 nccl=0"
    body = out[2:-2]
    assert len(body) == 2
    assert body[0].startswith("recv off=0 count=%d dtype=7 peer=0 comm=0x1234 stream=0x5678" % (6 * 5))
    assert body[1].startswith("recv off=%d count=%d dtype=7 peer=3 " % (6 * 12, 6 * 3))
    # a sending rank: one send of its rows to the root, nothing else
    out = _gather_drive(d, 3, 4, 2, -1, *counts)
    assert out == ["group_start", "send off=0 count=18 dtype=7 peer=2 comm=0x1234 stream=0x5678", "group_end", "rc=0 nccl=0"]
    # a rank without rows opens and closes an empty group (every rank of the communicator takes part in the group call)
    assert _gather_drive(d, 1, 4, 2, -1, *counts) == ["group_start", "group_end", "rc=0 nccl=0"]
    # root 0, failures: the first recv fails -> no further recv, group_end still reached, the ncclResult is reported
    out = _gather_drive(d, 0, 3, 0, 1, 4, 4, 4)
    assert [l.split()[0] for l in out] == ["copy", "group_start", "recv", "group_end", "rc=3"] and out[-1] == "rc=3 nccl=5"
    # group_end itself fails
    assert _gather_drive(d, 1, 3, 0, 100, 4, 4, 4)[-1] == "rc=3 nccl=3"
    # group_start fails: nothing is enqueued and no group is left open (none was opened)
    assert _gather_drive(d, 1, 3, 0, 0, 4, 4, 4) == ["group_start", "rc=2 nccl=2"]
    # the root's own copy fails before any group call
    assert _gather_drive(d, 0, 3, 0, 200, 4, 4, 4)[1:] == ["rc=1 nccl=0"]
    # the one-rank pre-flight (ppp_gather_self_loop): send to self + recv from self inside ONE group, no copy; the group is closed when the send fails
    assert _gather_drive(d, "self", -1, 9) == ["group_start", "send off=0 count=54 dtype=7 peer=0 comm=0x1234 stream=0x5678",
                                               "recv off=0 count=54 dtype=7 peer=0 comm=0x1234 stream=0x5678", "group_end", "rc=0 nccl=0"]
    assert _gather_drive(d, "self", 1, 9) == ["group_start", "send off=0 count=54 dtype=7 peer=0 comm=0x1234 stream=0x5678", "group_end", "rc=3 nccl=5"]
    assert _gather_drive(d, "self", -1, 0) == ["rc=0 nccl=0"]


def test_bench_refuses_to_start_ranks_from_under_a_profiler():
    """`bench.py --gpus N` starts its own ranks -- but never from a process a profiler's preload has already
    given a GPU context (rocprofv3 sets ROCP_* / ROCPROFILER_* and LD_PRELOAD for its child)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/librocprofiler-sdk-tool.so"
    env["PPP_BENCH_ECHO_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "under a profiler" in out and "rank 0 of 2" not in out
