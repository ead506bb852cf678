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
    if rank == 0:
        full = concat_robot_path(blocks)
        q.put(([b.shape[0] for b in blocks], full.numpy()))
    else:
        assert blocks is None
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


def test_gather_without_process_group():
    import torch
    from polishpathplanning_amd.robot_path import gather_robot_path
    t = torch.ones((3, 6))
    assert gather_robot_path(t)[0] is t
