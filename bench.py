#!/usr/bin/env python3
"""Headline benchmark: polishing waypoints/s on the BASELINE.json configuration.

One "step" = one full pass of the hot path (GenPath + getPath: a2..a15 of SURVEY.md section 8)
over one synthetic workpiece cloud that is already resident in HBM, ending with the finished
WayPointsList assembled on rank 0.  N = 1 runs configs[1] (1M-point wavy plate, 256 slices).
N > 1 shards a batch of workpieces one per GPU (weak scaling, no data-path collective) and
gathers the per-GPU robot paths to rank 0 over RCCL inside the timed region.
The K steps are enqueued back to back (each into one of two alternating output / gather buffers, ordered against the
collective's stream by events); the host synchronises once, at the end of the timed region, which is bracketed by
barrier + device synchronisation on both sides.
--mode slices (SURVEY.md 8e case ii, meant for cfg5_10m_s1024) instead shards the SLICES of one cloud:
GPU g plans slices [g*S/N, (g+1)*S/N), the pre-smoothing blocks are gathered to rank 0, which runs
postion_smooth / reduceRPY / flange offset once over the whole list (strong scaling).

Launch: python bench.py --gpus N --steps K --warmup W     (N > 1 via torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md "HBM3E peak BW")
HBM_MEASURED_COPY_GBS = 6290.0  # float4 copy measured on MI355X (same guide); SURVEY.md 8(d) asks for the fraction against both


def load_traffic(kernel, launches_per_pass, config):
    """HBM bytes per pass of `kernel` from the newest committed PMC summary (profiles/*_traffic.json,
    written by tools/collect_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled
    as MI355X_MICROARCH.md prescribes for gfx950).  None when no summary for this workload exists."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", "cfg2_1m_s256") == config and kernel in d.get("kernels", {}):
            best = (d["kernels"][kernel]["hbm_bytes_per_launch"] * launches_per_pass, os.path.basename(f))
    return best if best else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2_1m_s256")
    ap.add_argument("--batch", type=int, default=1,
                    help="workpieces per GPU per step, one engine handle (= one HIP stream) each (BASELINE config 3)")
    ap.add_argument("--mode", choices=["workpieces", "slices"], default="workpieces",
                    help="N > 1: one workpiece per GPU (weak scaling, default) or the slice ranges of ONE cloud (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-passes", type=int, default=20)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from polishpathplanning_amd import engine, synth
    from polishpathplanning_amd.robot_path import exchange_counts, gather_robot_path

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = world == 1 and os.environ.get("PPP_BENCH_FORCE_DIST") == "1"   # rehearsal: a one-rank RCCL group on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29571"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    if args.mode == "slices" and world > 1:
        return bench_slices(args, rank, local_rank, world, dev, dist, torch, engine, synth)

    # ---- synthetic workpiece of this rank (untimed: generation + H2D) ----
    base_seed = sorted(synth.CONFIGS).index(args.config) + 1
    engines, clouds = [], []
    for bi in range(args.batch):
        pts, cfg = synth.make_config(args.config, seed=base_seed + 1000 * rank + 17 * bi)
        e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
        e.set_cloud(pts)
        engines.append(e)
        clouds.append(pts)
    eng, pts = engines[0], clouds[0]
    n_points = int(pts.shape[0])

    # first pass: learn S and W, size the gather buffer
    S = eng.gen_path()
    W = eng.get_path()
    if cfg.get("slices") and S != cfg["slices"]:
        raise SystemExit("config %s produced %d slices, expected %d" % (args.config, S, cfg["slices"]))
    w_all = [W]
    for e in engines[1:]:
        e.gen_path()
        w_all.append(e.get_path())
    # the batch is fixed, so is every rank's waypoint count: exchanged once; the engines write straight into the
    # gatherer's send buffer, so a step's exchange is one collective and no copy.  N > 1 uses TWO gatherers: the
    # RCCL gather of step k-1 runs (on the framework's stream) while the planner works on step k (on its own stream).
    from polishpathplanning_amd.robot_path import RobotPathGatherer, StreamOrder, run_chained_steps, run_pipelined_steps, run_streamed_steps
    host_waits = os.environ.get("PPP_BENCH_HOST_WAITS") == "1"   # the earlier loop: one host wait per step (kept for comparison)
    gatherers = [RobotPathGatherer(sum(w_all), dist if (world > 1 or force_dist) else None, dev, force_collective=force_dist) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(w_all)[:-1]]).astype(np.int64)
    w_step = int(sum(w_all))
    # the planner's own stream, wrapped so that framework events can order it against the collective's stream
    # (one GPU, no collective: the "gather" is the identity on a buffer nobody else reads, there is nothing to order)
    from polishpathplanning_amd.robot_path import NoOrder
    planner_stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev)
    order = StreamOrder(torch, planner_stream) if gatherers[0].dist else NoOrder()

    def plan(k):
        # GenPath + getPath of every workpiece of this rank as ONE hipGraph launch (a branch per workpiece); every
        # branch ends by writing its WayPointsList to its place in the gather buffer
        engine.run_batch_async(engines, gatherers[k % 2].send.data_ptr(), offs, w_all)

    def run_steps(count):
        """`count` full steps: every step's robot path is planned and gathered on rank 0 before this returns.
        Steps alternate between two send/receive buffer pairs.  Step k's collective is enqueued behind step k's
        planning in the planner's own stream order, asynchronously, and is waited for (by the stream, not the host) two
        steps later, before its buffers are planned into again -- so it overlaps the planning of step k+1 and the host
        never waits inside the loop: it synchronises once, here, at the end (a failed step is reported by sync_batch)."""
        if count <= 0:
            return None
        if host_waits:
            return run_pipelined_steps(count, plan, lambda: engine.sync_batch(engines), gatherers,
                                       lambda: torch.cuda.current_stream().synchronize())
        if os.environ.get("PPP_BENCH_EVENT_ORDER") == "1":          # the collective on the framework's stream, events in between
            blocks = run_streamed_steps(count, plan, gatherers, order)
        else:
            blocks = run_chained_steps(count, plan, gatherers, planner_stream if gatherers[0].dist else None)
        engine.sync_batch(engines)
        torch.cuda.current_stream().synchronize()
        return blocks

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    blocks = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    w_local = w_step

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ww = torch.tensor([float(w_local)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ww, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    w_total = float(ww.item())
    value = w_total * args.steps / elapsed

    out = None
    if rank == 0:
        # ---- the last step's assembled robot path (untimed check): rank 0's own block is its engine's list, every
        # rank's block has the row count that rank announced ----
        assembled = None
        if blocks is not None:
            own = eng.waypoints()
            got0 = blocks[0][: own.shape[0]].cpu().numpy()
            assembled = {"rank0_block_equals_its_list": bool(np.array_equal(got0, own)),
                         "rows_per_rank": [int(b.shape[0]) for b in blocks]}
        # ---- roofline of the dominant kernel: HIP events on the engine's own stream ----
        eng.enable_timing(True)
        acc, launches = {}, {}
        for _ in range(args.profile_passes):
            eng.gen_path_async()
            eng.get_path_async()
            eng.sync()
            kt, kl = eng.kernel_times(with_launches=True)
            for k, v in kt.items():
                acc.setdefault(k, []).append(v)
            launches = kl
        eng.enable_timing(False)
        kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}  # per pass, summed over that kernel's launches
        dom = max(kern_ms, key=kern_ms.get)
        w_one = float(w_all[0])
        alg_bytes = 12.0 * n_points + 24.0 * w_one  # SURVEY.md 8(d), per workpiece: xyz read once + waypoints written once
        achieved = alg_bytes / (kern_ms[dom] * 1e-3) / 1e9
        traffic, traffic_src = load_traffic(dom, launches.get(dom, 1), args.config)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_MEASURED_COPY_GBS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "launches_per_pass": launches.get(dom, 1),
                    "avg_launch_ms": kern_ms[dom] / max(1, launches.get(dom, 1)),
                    "note": "one pass = one workpiece; kernel_ms are per pass (summed over a kernel's launches). "
                            "12.6 MB of algorithmic traffic is 2 us at the HBM roof: this workload is launch/latency bound",
                    "kernel_ms": {k: round(v, 5) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])},
                    "algorithmic_bytes": alg_bytes,
                    "pipeline_gbs": alg_bytes * args.batch / (elapsed / args.steps) / 1e9}

        # ---- CPU baseline: the oracle in reference-complexity mode on the node's host cores ----
        cpu = None
        err = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ppo
            o = ppo.Oracle(pts, tool_radius=cfg["tool_radius"], reference_complexity=1)
            t1 = time.perf_counter()
            o.gen_path()
            wo = o.get_path()
            t_cpu = time.perf_counter() - t1
            cpu = {"value": wo / t_cpu, "unit": "waypoints/s", "cores": 1, "kind": "port",
                   "sample": "1 full pass of %s (GenPath + getPath), single thread, reference-complexity "
                             "mode: per-slice O(N) PassThrough scans, whole-cloud normal estimation twice" % args.config,
                   "seconds": t_cpu, "host_cores_available": os.cpu_count()}
            try:   # the host the baseline ran on
                info = open("/proc/cpuinfo").read()
                models = [ln.split(":", 1)[1].strip() for ln in info.splitlines() if ln.startswith("model name")]
                sockets = {ln.split(":", 1)[1].strip() for ln in info.splitlines() if ln.startswith("physical id")}
                cpu["cpu_model"] = models[0] if models else None
                cpu["sockets"] = len(sockets) or None
            except OSError:
                pass
            # context only: the same pass with OpenMP over the slices and over the points of the normal
            # estimation; the kd-tree builds stay serial, as FLANN's are in the reference
            try:
                nt = len(os.sched_getaffinity(0))
            except AttributeError:
                nt = os.cpu_count() or 1
            nt = min(nt, 16)   # the CPU share of one GPU on the bench node
            if nt > 1:
                o2 = ppo.Oracle(pts, tool_radius=cfg["tool_radius"], reference_complexity=1, threads=nt)
                t2 = time.perf_counter()
                o2.gen_path()
                wo2 = o2.get_path()
                t_all = time.perf_counter() - t2
                cpu["all_cores"] = {"value": wo2 / t_all, "cores": nt, "seconds": t_all,
                                    "note": "not the reference's behaviour (its hot path is single-threaded): context only"}
            gw = eng.waypoints()
            ow = o.waypoints()
            if gw.shape == ow.shape and len(gw):
                d = np.linalg.norm(gw[:, :3] - ow[:, :3], axis=1)
                err = {"max_m": float(d.max()), "rms_m": float(np.sqrt((d ** 2).mean())), "waypoints_equal": True}
            else:
                err = {"waypoints_equal": False, "gpu": int(gw.shape[0]), "oracle": int(ow.shape[0])}

        out = {
            "metric": "polishing waypoints/sec for 1M-pt cloud, 256 slices; path L2 err vs ref",
            "value": value, "unit": "waypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "points_per_workpiece": n_points, "slices": S,
                       "waypoints_per_workpiece": int(w_all[0]), "workpieces": world * args.batch, "batch_per_gpu": args.batch,
                       "parallelism": "one workpiece per GPU, RCCL gather of robot_path to rank 0" if world > 1 else "single GPU",
                       "pairing": "kd", "walk": "center_int (connect)", "tool_radius_mm": cfg["tool_radius"]},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "path_l2_err": err,
            "assembled_path": assembled,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def bench_slices(args, rank, local_rank, world, dev, dist, torch, engine, synth):
    """One cloud, slices sharded over the GPUs (no data-path collective until the gather); rank 0 finishes the list."""
    from polishpathplanning_amd.robot_path import exchange_counts, gather_robot_path, slice_ranges
    base_seed = sorted(synth.CONFIGS).index(args.config) + 1
    pts, cfg = synth.make_config(args.config, seed=base_seed)      # every rank: the same cloud
    probe = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
    probe.set_cloud(pts)
    S = len(probe.slice_positions())
    probe.close()
    b, e = slice_ranges(S, world)[rank]
    eng = None
    nkept = max(S - 2, 0)
    counts_local = np.zeros(nkept, np.int32)
    w_local = 0
    if b < e:
        eng = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], slice_begin=b, slice_end=e)
        eng.set_cloud(pts)
        eng.gen_path()
        w_local = eng.get_path()
        counts_local = eng.waypoint_counts()
    elif rank == 0:          # more GPUs than slices: rank 0 still needs a handle, it finishes the list
        eng = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], slice_begin=0, slice_end=1)
        eng.set_cloud(pts)
    cnt = torch.from_numpy(counts_local.astype(np.int64)).to(dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    counts_all = cnt.cpu().numpy().astype(np.int32)                  # fixed workload: exchanged once
    w_ranks = exchange_counts(w_local, dist, dev)
    W = int(counts_all.sum())
    send = torch.zeros((max(w_local, 1), 6), dtype=torch.float32, device=dev)

    def step():
        w = 0
        if eng is not None and b < e:
            eng.run_async()                                          # a2..a12 of this rank's slices, one hipGraph launch
            w = eng.copy_stage_to_device(engine.STAGE_WP_PRESMOOTH, send.data_ptr(), send.shape[0])
        blocks = gather_robot_path(send[:w], dist, dev, w_ranks)
        if rank == 0:
            pre = torch.cat(blocks, dim=0).contiguous()
            eng.finish_path_async(pre.data_ptr(), pre.shape[0], counts_all)   # a13..a15 once, over the whole list
            eng.sync()
            return pre.shape[0]
        return 0

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = step()
    fence()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    out = None
    if rank == 0:
        assert got == W
        n_points = int(pts.shape[0])
        alg_bytes = 12.0 * n_points + 24.0 * W
        out = {
            "metric": "polishing waypoints/sec for 1M-pt cloud, 256 slices; path L2 err vs ref",
            "value": W * args.steps / elapsed, "unit": "waypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "points_per_workpiece": n_points, "slices": S, "waypoints_per_workpiece": W,
                       "workpieces": 1, "parallelism": "slice ranges of one cloud, one range per GPU; RCCL gather of the pre-smoothing "
                       "blocks; postion_smooth/reduceRPY/flange once on rank 0", "tool_radius_mm": cfg["tool_radius"]},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": alg_bytes / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole pipeline, every rank reads the whole cloud once (k_minmax) and indexes its own x interval"},
            "cpu_baseline": None,
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
