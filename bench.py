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
        ks = d.get("kernels", {})
        if d.get("workload", "cfg2_1m_s256") != config:
            continue
        if kernel in ks:
            best = (ks[kernel]["hbm_bytes_per_launch"] * launches_per_pass, os.path.basename(f))
        else:   # one kernel launched with several grids (a batch as two halves in the timed loop, full width under the event timers):
            #     the summary keeps them apart as <kernel>@grid<threads>; the roofline launch is the widest one
            wide = sorted((int(k.split("@grid")[1]), k) for k in ks if k.startswith(kernel + "@grid") and k.split("@grid")[1].isdigit())
            if wide:
                best = (ks[wide[-1][1]]["hbm_bytes_per_launch"] * launches_per_pass, os.path.basename(f) + " (" + wide[-1][1] + ")")
    return best if best else (None, None)


def load_traffic_total(launches, config):
    """HBM bytes of one whole pass: the sum of load_traffic over every kernel of the pass (None when a kernel has no committed summary)."""
    total, srcs = 0.0, set()
    for k, n in launches.items():
        t, src = load_traffic(k, n, config)
        if t is None:
            return None, None
        total += t
        srcs.add(src.split(" (")[0])
    return (total, ", ".join(sorted(srcs))) if launches else (None, None)


_RECORD_FD = None


def emit_record(out):
    """the one line of the run, on the process's real stdout"""
    line = (json.dumps(out) + "\n").encode()
    sys.stdout.flush()
    if _RECORD_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        while line:
            line = line[os.write(_RECORD_FD, line):]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None,
                    help="workload (polishpathplanning_amd.synth.CONFIGS); default: cfg2_1m_s256 on one GPU (BASELINE configs[1]), "
                         "cfg4_2m_s256 per GPU for --gpus N > 1 (configs[3]), cfg5_10m_s1024 for --mode slices (configs[4])")
    ap.add_argument("--batch", type=int, default=1,
                    help="workpieces per GPU per step, one engine handle (= one HIP stream) each (BASELINE config 3)")
    ap.add_argument("--mode", choices=["workpieces", "slices"], default="workpieces",
                    help="N > 1: one workpiece per GPU (weak scaling, default) or the slice ranges of ONE cloud (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-passes", type=int, default=20)
    ap.add_argument("--no-single-handle", action="store_true", help="skip the one-handle loop that is timed beside the steps taking turns (profiling runs)")
    ap.add_argument("--handles", type=int, default=0,
                    help="engine handles (= HIP streams) the steps of a one-GPU, one-workpiece run take turns on; 0 = 3 where that applies, else 1")
    ap.add_argument("--rotate", type=int, default=None,
                    help="also report the step time over K distinct resident clouds planned round-robin (K x working set beyond "
                         "the 256 MiB Infinity Cache) and the cold time of a never-seen cloud; 0 = off, the default run uses 8")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the Dynamic_adjustment = true measurement that the default one-GPU run reports beside the headline")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the cfg3 x 64 batch and the cfg5 measurement that the default one-GPU run reports beside the headline")
    ap.add_argument("--dynamic", action="store_true",
                    help="plan with Dynamic_adjustment = true (the reference's config.txt default; SURVEY.md 8f rank 1): the same "
                         "workload through the slice-to-slice chains -- a measurement beside the headline, not the headline")
    args = ap.parse_args()
    if args.rotate is None:
        args.rotate = 8 if (args.gpus == 1 and args.batch == 1 and not args.dynamic) else 0
    if args.config is None:
        args.config = "cfg5_10m_s1024" if (args.mode == "slices" and args.gpus > 1) else ("cfg4_2m_s256" if args.gpus > 1 else "cfg2_1m_s256")

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE anything touches the GPU (a
    # process that has initialised HIP must never exec or fork GPU children), relay rank 0's JSON line and exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # ... and never from under a profiler: its preloaded library has initialised the GPU in THIS process already, so starting
        # the launcher from here would be exactly the hop this pool forbids (profile the one-rank rehearsal instead)
        preload = os.environ.get("LD_PRELOAD", "")
        if "rocprof" in preload or any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in os.environ):
            raise SystemExit("bench.py --gpus %d under a profiler: start the ranks with torch.distributed.run yourself, or profile "
                             "the single-rank rehearsal (PPP_BENCH_FORCE_DIST=1 --gpus 1)" % args.gpus)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # stdout carries ONE line, the JSON record: whatever native libraries print while they come up (RCCL's version banner goes to
    # stdout) is sent to stderr; the record is written to the saved descriptor at the end
    sys.stdout.flush()
    global _RECORD_FD
    _RECORD_FD = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from polishpathplanning_amd import engine, synth
    from polishpathplanning_amd.robot_path import exchange_counts, gather_robot_path

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("PPP_BENCH_ECHO_RANK") == "1":
        print("bench.py rank %d of %d: %s, mode %s" % (rank, world, args.config, args.mode), file=sys.stderr, flush=True)
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

    if args.mode == "slices" and (world > 1 or force_dist):
        return bench_slices(args, rank, local_rank, world, dev, dist, torch, engine, synth)

    # ---- synthetic workpiece of this rank (untimed: generation + H2D) ----
    base_seed = sorted(synth.CONFIGS).index(args.config) + 1
    engines, clouds = [], []
    amp_rng = np.random.default_rng(base_seed + 1000 * rank)
    for bi in range(args.batch):
        over = {}
        if args.batch > 1:   # every workpiece of a batch has its own seed and surface amplitude
            over["amp"] = float(synth.CONFIGS[args.config]["amp"] * amp_rng.uniform(0.5, 1.5))
        pts, cfg = synth.make_config(args.config, seed=base_seed + 1000 * rank + 17 * bi, **over)
        e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], dynamic_adjustment=1 if args.dynamic else 0)
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

    # ---- one GPU, one workpiece per step: consecutive steps are independent workpieces, so they take turns on H handles.
    # A handle is a HIP stream with its own buffers: step k's binning launch runs beside step k-1's per-slice kernel instead
    # of behind its finish launch (three dependent launches per pass leave most of the chip idle most of the time).  Every
    # replica holds the same resident cloud and plans the whole pass; each writes its list to a buffer of its own.
    # With an exchange in the step (N > 1, the one-rank rehearsal) every handle has a buffer pair and a planner stream of its own; the
    # framework runs the collectives one after the other on the communicator's stream, in the order they are called (run_chained_steps).
    turns = args.handles if args.handles > 0 else 3
    with_exchange = world > 1 or force_dist
    if args.batch != 1 or args.dynamic or host_waits or os.environ.get("PPP_BENCH_EVENT_ORDER") == "1":
        turns = 1

    replicas, outs = [eng], []
    if turns > 1:
        # (the first handle takes part: a fourth stream in the process -- even an idle one -- shares a hardware queue with one of the
        #  three, 0.039 -> 0.048 ms per step; it is told about its neighbours for these loops only, see side_by_side below)
        for _ in range(turns - 1):
            e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
            e.set_side_by_side(turns)
            e.set_cloud(pts)
            if (e.gen_path(), e.get_path()) != (S, W):
                raise SystemExit("a replica of the workpiece planned another list")
            replicas.append(e)
        if with_exchange:   # a buffer pair and a planner stream per handle; the collectives run in call order on the communicator's own stream
            turn_gatherers = [RobotPathGatherer(sum(w_all), dist, dev, force_collective=force_dist) for _ in range(turns)]
            outs = [g.send for g in turn_gatherers]
            turn_streams = [torch.cuda.ExternalStream(r.stream_ptr(), device=dev) for r in replicas]
        else:
            outs = [torch.empty((max(W, 1), 6), dtype=torch.float32, device=dev) for _ in range(turns)]

    def plan_turn(k):
        engine.run_batch_async([replicas[k % turns]], outs[k % turns].data_ptr(), offs, w_all)

    def run_turns(count):
        if count <= 0:
            return None
        if with_exchange:
            blocks = run_chained_steps(count, plan_turn, turn_gatherers, turn_streams)
            engine.sync_batch(replicas)
            torch.cuda.current_stream().synchronize()
            return blocks
        for k in range(count):
            plan_turn(k)
        engine.sync_batch(replicas)
        torch.cuda.current_stream().synchronize()
        return [outs[(count - 1) % turns][:W]]

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

    def side_by_side(n):
        # how many passes share the device: from two on the plan keeps the slice workgroups of small windows at 512 threads, which leaves
        # room on a CU for a neighbouring pass's binning and finish workgroups (ppp_set_side_by_side); the first handle is switched back
        # for the one-handle loop and for the launches' durations alone on the device
        if turns > 1:
            eng.set_side_by_side(n)

    timed = run_turns if turns > 1 else run_steps
    side_by_side(turns)
    timed(args.warmup)
    fence()
    t0 = time.perf_counter()
    blocks = timed(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    w_local = w_step
    single = None
    side_by_side(1)
    if turns > 1 and not args.no_single_handle:   # the same K steps on ONE handle, back to back (rounds 1-3's loop), for the record
        run_steps(args.warmup)
        fence()
        t1 = time.perf_counter()
        run_steps(args.steps)
        fence()
        e1 = time.perf_counter() - t1
        single = {"ms_per_step": e1 / args.steps * 1e3, "value": w_step * args.steps / e1,
                  "replicas_equal": bool(all(np.array_equal(r.waypoints(), eng.waypoints()) for r in replicas[1:])),
                  "note": "the same steps enqueued back to back on one handle (one stream): what rounds 1-3 report as the headline"}

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ww = torch.tensor([float(w_local)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ww, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    w_total = float(ww.item())
    value = w_total * args.steps / elapsed

    # ---- N > 1: how many ranks took part (as torch.distributed and as the RCCL communicator itself see it), every rank's
    # waypoint count, and what the gather alone costs per step (untimed, after the timed region) ----
    multi = None
    if world > 1 or force_dist:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # every rank of the communicator adds one
        wl = torch.zeros(max(world, 1), dtype=torch.float64, device=dev)
        wl[rank] = float(w_local)
        dist.all_reduce(wl, op=dist.ReduceOp.SUM)
        reps = max(5, min(args.steps, 20))
        torch.cuda.synchronize(); dist.barrier()
        tg0 = time.perf_counter()
        for _ in range(reps):
            gatherers[0].gather()
        torch.cuda.synchronize(); dist.barrier()
        tg = torch.tensor([(time.perf_counter() - tg0) / reps * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        engine_gather = None
        if force_dist and args.batch == 1:
            # the engine's own exchange (ppp_gather_waypoints: ncclSend / ncclRecv group on the PLANNER's stream) rehearsed against the real
            # librccl with a one-rank communicator -- the block sent to itself and received -- beside the framework's
            # gather above: step time with and without it, same handle, same stream, host wait once per loop
            try:
                from polishpathplanning_amd.robot_path import RcclComm
                comm = RcclComm(0, 1)
                rbuf = torch.empty((max(w_all[0], 1), 6), dtype=torch.float32, device=dev)

                def loop(with_gather, count):
                    t = time.perf_counter()
                    for _ in range(count):
                        eng.run_async()
                        if with_gather:
                            eng.gather_waypoints(comm.ptr, 0, 1, 0, [w_all[0]], rbuf.data_ptr())
                    eng.sync()
                    return (time.perf_counter() - t) / count * 1e3
                cnt = max(10, args.steps)
                loop(True, 3); loop(False, 3)
                t_with = min(loop(True, cnt) for _ in range(3)); t_without = min(loop(False, cnt) for _ in range(3))
                same = bool(np.array_equal(rbuf[: w_all[0]].cpu().numpy(), eng.waypoints()))
                engine_gather = {"step_ms_with_group_send_recv": t_with, "step_ms_without_exchange": t_without, "added_us": (t_with - t_without) * 1e3,
                                 "received_block_equals_the_list": same,
                                 "note": "one rank, real librccl: ncclSend to self + ncclRecv from self in one group on the planner's stream"}
                comm.close()
            except Exception as ex:
                engine_gather = {"error": "%s: %s" % (type(ex).__name__, ex)}
        multi = {"n_ranks_seen": {"torch_distributed_world_size": int(dist.get_world_size()), "rccl_allreduce_of_ones": int(round(float(ones.item())))},
                 "engine_gather_rehearsal": engine_gather,
                 "waypoints_per_rank": [int(round(float(x))) for x in wl.cpu().tolist()],
                 "gather_ms_per_step_alone": float(tg.item()),
                 "note": "gather_ms_per_step_alone: the collective back to back with a host wait around the loop; inside the timed loop it "
                         "overlaps the next step's planning"}

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
            # the launches of the timed loop (ppp_run_batch_async: one launch per stage over the step's workpieces), enqueued
            # directly instead of as a graph replay so that HIP events on the engine's own stream bracket each of them
            engine.run_batch_async(engines, gatherers[0].send.data_ptr(), offs, w_all)
            engine.sync_batch(engines)
            kt, kl = eng.kernel_times(with_launches=True)
            for k, v in kt.items():
                acc.setdefault(k, []).append(v)
            launches = kl
        eng.enable_timing(False)
        kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}  # per pass, summed over that kernel's launches
        dom = max(kern_ms, key=kern_ms.get)
        # ... and the same launches while the steps take turns on the handles of the timed loop: a launch then shares the chip with
        # the launches of the neighbouring steps, its own wall time (HIP events on ITS stream) grows while the loop's throughput does too
        kern_ms_turns = None
        if turns > 1:
            side_by_side(turns)
            for r in replicas:
                r.enable_timing(True)
            acc2 = {}
            rounds = max(2, args.profile_passes // 2)
            for _ in range(rounds):
                for k in range(2 * turns):
                    engine.run_batch_async([replicas[k % turns]], outs[k % turns].data_ptr(), offs, w_all)
                engine.sync_batch(replicas)
                for r in replicas:
                    kt2, kl2 = r.kernel_times(with_launches=True)   # (summed over the passes this handle ran since the last call)
                    for k2, v2 in kt2.items():
                        acc2.setdefault(k2, []).append(v2 / 2.0)
            for r in replicas:
                r.enable_timing(False)
            side_by_side(1)
            kern_ms_turns = {k2: float(np.mean(v2)) for k2, v2 in acc2.items()}
        # 12 B per point read once + 24 B per waypoint written once; one launch processes the whole batch
        alg_bytes = 12.0 * float(sum(int(c.shape[0]) for c in clouds)) + 24.0 * float(sum(w_all))
        achieved = alg_bytes / (kern_ms[dom] * 1e-3) / 1e9
        workload_key = args.config + ("_b%d" % args.batch if args.batch > 1 else "") + ("_dyn" if args.dynamic else "")
        traffic, traffic_src = load_traffic(dom, launches.get(dom, 1), workload_key)
        roofline = {"bound": "hbm", "kernel": dom, "launch_path": ("window (%d launches)" if eng.fast_path() else "slab index (%d launches)") % sum(launches.values()),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_MEASURED_COPY_GBS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_total": load_traffic_total(launches, workload_key)[0],
                    "launches_per_pass": launches.get(dom, 1),
                    "avg_launch_ms": kern_ms[dom] / max(1, launches.get(dom, 1)),
                    "avg_launch_ms_steps_taking_turns": (kern_ms_turns.get(dom) / max(1, launches.get(dom, 1))) if kern_ms_turns and dom in kern_ms_turns else None,
                    "frac_steps_taking_turns": (alg_bytes / (kern_ms_turns[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern_ms_turns and dom in kern_ms_turns else None,
                    "kernel_ms_steps_taking_turns": ({k: round(v, 5) for k, v in sorted(kern_ms_turns.items(), key=lambda kv: -kv[1])} if kern_ms_turns else None),
                    "pipeline_frac": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "note": "one pass = one launch of every stage over the %d workpiece(s) of a step; kernel_ms are per pass (summed "
                            "over a kernel's launches), a launch alone on the device (achieved / frac / avg_launch_ms: as in rounds 1-3); "
                            "*_steps_taking_turns: the same launches measured while consecutive steps overlap on the timed loop's handles "
                            "(a launch's own duration grows, the loop's throughput -- pipeline_gbs, pipeline_frac -- grows too); "
                            "algorithmic_bytes = 12 B x points + 24 B x waypoints of the step; traffic = HBM bytes "
                            "of the dominant kernel's launches, traffic_total = of every launch of the pass (committed PMC summaries)" % args.batch,
                    "kernel_ms": {k: round(v, 5) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])},
                    "algorithmic_bytes": alg_bytes,
                    "pipeline_gbs": alg_bytes / (elapsed / args.steps) / 1e9}

        # ---- what a stream of NEW workpieces sees: the replay above plans one resident cloud whose working
        # set sits in the 256 MiB Infinity Cache, with its slab grid and buffers prepared by set_cloud ----
        latency = None
        if world == 1 and args.batch == 1 and args.rotate > 0:
            latency = measure_rotation_and_cold(args, torch, dev, engine, synth, cfg, base_seed, eng, pts, local_rank)

        # ---- CPU baseline: the oracle in reference-complexity mode on the node's host cores ----
        cpu = None
        err = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ppo
            o = ppo.Oracle(pts, tool_radius=cfg["tool_radius"], reference_complexity=1, dynamic_adjustment=1 if args.dynamic else 0)
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
                o2 = ppo.Oracle(pts, tool_radius=cfg["tool_radius"], reference_complexity=1, threads=nt, dynamic_adjustment=1 if args.dynamic else 0)
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

        # ---- the reference's DEFAULT mode (config.txt:13 Dynamic_adjustment = true) on the same cloud: a second line of the
        # record, not the headline (the slice-to-slice chains are a launch chain) ----
        dynamic = None
        if world == 1 and args.batch == 1 and not args.dynamic and not args.no_dynamic:
            dynamic = measure_dynamic(args, engine, cfg, pts, local_rank, None if args.no_cpu_baseline else "oracle")

        # ---- the bandwidth-sized BASELINE configurations (configs[2]: 64 x 250 k points in batched launches; configs[4]: 10 M points /
        # 1024 slices) in the same process, after the headline: lines of the record, never the headline ----
        other = None
        if world == 1 and args.batch == 1 and not args.dynamic and not args.no_other_configs and args.config == "cfg2_1m_s256":
            other = {}
            for key, name, nb, checked in (("cfg3b64", "cfg3_250k_s128", 64, 16), ("cfg5", "cfg5_10m_s1024", 1, 1)):
                try:
                    other[key] = measure_other_config(args, engine, synth, local_rank, name, nb, 0 if args.no_cpu_baseline else checked)
                except Exception as ex:   # the headline line must not depend on these
                    other[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        out = {
            "metric": "polishing waypoints/sec for 1M-pt cloud, 256 slices; path L2 err vs ref",
            "value": value, "unit": "waypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "points_per_workpiece": n_points, "slices": S,
                       "waypoints_per_workpiece": int(w_all[0]), "workpieces": world * args.batch, "batch_per_gpu": args.batch,
                       "parallelism": "one workpiece per GPU, RCCL gather of robot_path to rank 0" if world > 1 else "single GPU",
                       "pairing": "kd", "walk": "center_int (connect)", "tool_radius_mm": cfg["tool_radius"],
                       "dynamic_adjustment": bool(args.dynamic),
                       "handles_taking_turns": turns,
                       "step_order": ("consecutive steps (independent workpieces) take turns on %d engine handles = %d HIP streams, each with the "
                                      "same resident cloud and buffers of its own: a step's binning launch runs beside the step before's per-slice "
                                      "kernel; K steps enqueued, one host wait at the end" % (turns, turns)) if turns > 1 else
                                     "steps enqueued back to back on one handle (one HIP stream), one host wait at the end"},
            "single_handle": single,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "path_l2_err": err,
            "assembled_path": assembled,
            "latency": latency,
            "dynamic": dynamic,
            "other_configs": other,
            "multi_gpu": multi,
            "exchange": ("torch.distributed gather to rank 0 over RCCL (direct send/recv of padded blocks), asynchronous behind each "
                         "step in the planner's stream order" if gatherers[0].dist else None),
        }
        emit_record(out)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def measure_other_config(args, engine, synth, local_rank, name, nb, n_checked):
    """One of the bandwidth-sized BASELINE configurations through the same entry point as the headline's timed loop
    (ppp_run_batch_async: one launch per stage over the step's workpieces), clouds resident, `steps` steps enqueued back to back
    with one host wait at the end; then the stages' HIP-event durations, and the path error of `n_checked` members against the
    oracle's fast mode (its kd index instead of per-slice scans: a check, not a timed baseline)."""
    from polishpathplanning_amd.hipbuf import DeviceBuffer
    base_seed = sorted(synth.CONFIGS).index(name) + 1
    amp_rng = np.random.default_rng(base_seed)
    engines, clouds, w_all = [], [], []
    for bi in range(nb):
        over = {}
        if nb > 1:
            over["amp"] = float(synth.CONFIGS[name]["amp"] * amp_rng.uniform(0.5, 1.5))
        pts, cfg = synth.make_config(name, seed=base_seed + 17 * bi, **over)
        e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
        e.set_cloud(pts)
        S = e.gen_path()
        w_all.append(e.get_path())
        if cfg.get("slices") and S != cfg["slices"]:
            raise RuntimeError("config %s produced %d slices, expected %d" % (name, S, cfg["slices"]))
        engines.append(e); clouds.append(pts)
    offs = np.concatenate([[0], np.cumsum(w_all)[:-1]]).astype(np.int64)
    buf = DeviceBuffer(int(sum(w_all)) * 24)
    steps = max(5, min(args.steps, 20))
    for _ in range(3):
        engine.run_batch_async(engines, buf.ptr, offs, w_all)
    engine.sync_batch(engines)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(steps):
            engine.run_batch_async(engines, buf.ptr, offs, w_all)
        engine.sync_batch(engines)
        dt = (time.perf_counter() - t) / steps
        best = dt if best is None else min(best, dt)
    eng = engines[0]
    turns_ms = None
    if nb == 1:   # one workpiece per step: the steps take turns on three handles, as in the headline's loop
        reps, bufs = [eng], [buf]
        for _ in range(2):
            e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
            e.set_cloud(clouds[0]); e.gen_path(); e.get_path()
            reps.append(e); bufs.append(DeviceBuffer(int(sum(w_all)) * 24))
        for k in range(6):
            engine.run_batch_async([reps[k % 3]], bufs[k % 3].ptr, offs, w_all)
        engine.sync_batch(reps)
        for _ in range(3):
            t = time.perf_counter()
            for k in range(3 * steps):
                engine.run_batch_async([reps[k % 3]], bufs[k % 3].ptr, offs, w_all)
            engine.sync_batch(reps)
            dt = (time.perf_counter() - t) / (3 * steps)
            turns_ms = dt * 1e3 if turns_ms is None else min(turns_ms, dt * 1e3)
        for e in reps[1:]:
            e.close()
        for b in bufs[1:]:
            b.free()
    eng.enable_timing(True)
    acc, launches = {}, {}
    for _ in range(5):
        engine.run_batch_async(engines, buf.ptr, offs, w_all)
        engine.sync_batch(engines)
        kt, launches = eng.kernel_times(with_launches=True)
        for k, v in kt.items():
            acc.setdefault(k, []).append(v)
    eng.enable_timing(False)
    kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    dom = max(kern_ms, key=kern_ms.get)
    n_points = int(sum(int(c.shape[0]) for c in clouds))
    alg_bytes = 12.0 * n_points + 24.0 * float(sum(w_all))
    workload_key = name + ("_b%d" % nb if nb > 1 else "")
    out = {"workload": name, "workpieces": nb, "points": n_points, "slices_per_workpiece": int(engines[0].num_slices()), "waypoints": int(sum(w_all)),
           "steps": steps, "ms_per_step": best * 1e3, "waypoints_per_s": float(sum(w_all)) / best, "algorithmic_bytes": alg_bytes,
           "pipeline_gbs": alg_bytes / best / 1e9, "pipeline_frac_of_peak": alg_bytes / best / 1e9 / HBM_PEAK_GBS,
           "launch_path": ("window (%d launches)" if eng.fast_path() else "slab index (%d launches)") % sum(launches.values()),
           "dominant_kernel": dom, "dominant_kernel_ms": kern_ms[dom], "dominant_kernel_gbs": alg_bytes / (kern_ms[dom] * 1e-3) / 1e9,
           "dominant_kernel_frac": alg_bytes / (kern_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "kernel_ms": {k: round(v, 5) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])},
           "traffic_total": load_traffic_total(launches, workload_key)[0],
           "ms_per_step_three_handles": turns_ms, "waypoints_per_s_three_handles": (float(sum(w_all)) / (turns_ms * 1e-3)) if turns_ms else None,
           "note": "best of 3 runs of `steps` steps enqueued back to back, one host wait per run (ms_per_step: on one handle; ms_per_step_three_handles: "
                   "the steps taking turns on three handles with the same resident cloud, as in the headline's loop); kernel_ms from HIP events around every "
                   "launch of a pass enqueued directly (full-width launches; the timed loop's graph may run a batch as two halves)"}
    if n_checked:
        from oracle import ppo
        members = sorted(set(int(round(q)) for q in np.linspace(0, nb - 1, min(n_checked, nb))))
        worst, rms2, rows, equal = 0.0, 0.0, 0, True
        for bi in members:
            o = ppo.Oracle(clouds[bi], tool_radius=synth.CONFIGS[name]["tool_radius"])
            o.gen_path(); o.get_path()
            gw, ow = engines[bi].waypoints(), o.waypoints()
            if gw.shape != ow.shape or not len(gw):
                equal = False
                continue
            d = np.linalg.norm(gw[:, :3] - ow[:, :3], axis=1)
            worst = max(worst, float(d.max())); rms2 += float((d ** 2).sum()); rows += len(d)
        out["path_l2_err"] = {"max_m": worst, "rms_m": float(np.sqrt(rms2 / max(rows, 1))), "waypoints_equal": equal, "members_checked": members,
                              "against": "the oracle's fast mode"}
    for e in engines:
        e.close()
    return out


def measure_dynamic(args, engine, cfg, pts, local_rank, check):
    """The same cloud planned with Dynamic_adjustment = true (path_dynamic_alg.cpp:183-334): ms per step as a hipGraph replay,
    waypoints/s, and the path error against the oracle's own dynamic adjustment (the oracle with its kd index: a check, not a
    timed baseline)."""
    e = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], dynamic_adjustment=1)
    e.set_cloud(pts)
    e.gen_path()
    W = e.get_path()
    steps = max(3, min(args.steps, 10))
    for _ in range(2):
        e.run_async()
    e.sync()
    t = time.perf_counter()
    for _ in range(steps):
        e.run_async()
    e.sync()
    ms = (time.perf_counter() - t) / steps * 1e3
    # ... and as a stream of workpieces: the chain of one workpiece fills a quarter of the chip (one wave per Area2Cloud evaluation), the
    # chains of three workpieces on three handles run side by side
    reps = [e]
    for _ in range(2):
        r = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], dynamic_adjustment=1)
        r.set_cloud(pts); r.gen_path(); r.get_path()
        r.run_async(); r.sync()
        reps.append(r)
    ms3 = None
    for _ in range(2):
        t = time.perf_counter()
        for k in range(3 * steps):
            reps[k % 3].run_async()
        for r in reps:
            r.sync()
        dt = (time.perf_counter() - t) / (3 * steps) * 1e3
        ms3 = dt if ms3 is None else min(ms3, dt)
    same = bool(all(np.array_equal(r.waypoints(), e.waypoints()) for r in reps[1:]))
    for r in reps[1:]:
        r.close()
    out = {"ms_per_step": ms, "waypoints_per_s": W / (ms * 1e-3), "waypoints": int(W), "steps": steps,
           "ms_per_step_three_handles": ms3, "waypoints_per_s_three_handles": W / (ms3 * 1e-3), "replicas_equal": same,
           "note": "Dynamic_adjustment = true (the reference's config.txt default): slice s is re-fitted against the boundary of slice s-1, "
                   "so the slices form a chain of dependent launches; ms_per_step: one workpiece at a time on one handle; "
                   "ms_per_step_three_handles: consecutive workpieces taking turns on three handles (their chains run side by side)"}
    if check == "oracle":
        from oracle import ppo
        o = ppo.Oracle(pts, tool_radius=cfg["tool_radius"], dynamic_adjustment=1)
        o.gen_path()
        o.get_path()
        gw, ow = e.waypoints(), o.waypoints()
        if gw.shape == ow.shape and len(gw):
            d = np.linalg.norm(gw[:, :3] - ow[:, :3], axis=1)
            out["path_l2_err"] = {"max_m": float(d.max()), "rms_m": float(np.sqrt((d ** 2).mean())), "waypoints_equal": True}
        else:
            out["path_l2_err"] = {"waypoints_equal": False, "gpu": int(gw.shape[0]), "oracle": int(ow.shape[0])}
    e.close()
    return out


def measure_rotation_and_cold(args, torch, dev, engine, synth, cfg, base_seed, eng, pts, local_rank):
    """K distinct resident clouds planned round-robin (K x ~45 MB at cfg 2: beyond the Infinity Cache) against the same
    loop on ONE cloud, both with a host wait per step (the clouds live on different handles = different streams); and the
    cold path of a never-seen cloud: ppp_set_cloud_device (ingest, bounds, plan) + the first ppp_run_async (graph capture
    and instantiation included) + the wait for the list, cloud already in device memory."""
    K = args.rotate
    steps = max(args.steps, 2 * K)
    others = []
    for k in range(1, K):
        p2, _ = synth.make_config(args.config, seed=base_seed + 7919 * k)
        e2 = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
        e2.set_cloud(p2)
        e2.run_async(); e2.sync()
        others.append(e2)
    ring = [eng] + others

    def loop(engs, count):
        t = time.perf_counter()
        for k in range(count):
            e = engs[k % len(engs)]
            e.run_async()
            e.sync()
        return (time.perf_counter() - t) / count * 1e3

    eng.run_async(); eng.sync()
    loop([eng], 5); one = min(loop([eng], steps) for _ in range(3))
    loop(ring, K); rot = min(loop(ring, steps) for _ in range(3))
    for e2 in others:
        e2.close()
    # cold: new clouds of the same size into one handle
    cold = []
    ec = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
    d = None   # one device buffer for all of them: a fresh allocation's first touch (page tables, TLB) is the framework's cost, not the planner's
    for k in range(5):
        p2, _ = synth.make_config(args.config, seed=base_seed + 104729 * (k + 1))
        src = torch.from_numpy(np.ascontiguousarray(p2))
        if d is None or d.shape != src.shape:
            d = torch.empty_like(src, device=dev)
        d.copy_(src)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ec.set_cloud_device_async(d.data_ptr(), int(p2.shape[0]), 12)   # (d stays as it is until ec.sync() below has returned)
        t1 = time.perf_counter()
        ec.run_async()
        ec.sync()
        t2 = time.perf_counter()
        cold.append(((t2 - t) * 1e3, (t1 - t) * 1e3, (t2 - t1) * 1e3))
    ec.close()
    del d
    c_caller = cold_path_from_c(args, engine, synth, base_seed)
    return {"replay_one_cloud_hostwait_ms": one, "rotate_clouds": K, "rotate_hostwait_ms": rot,
            "cold_first_ms": cold[0][0], "cold_ms": min(c[0] for c in cold[1:]), "cold_median_ms": float(np.median([c[0] for c in cold[1:]])),
            "cold_split_ms": {"set_cloud_device": min(c[1] for c in cold[1:]), "first_run_async_and_wait": min(c[2] for c in cold[1:])},
            "cold_c_caller_ms": c_caller[0], "cold_c_caller_median_ms": c_caller[1],
            "new_clouds_through_queue_ms_by_lanes": c_caller[2],
            "note": "host wait after every step in both loops; cold = set_cloud_device_async + first run_async + wait on a never-seen cloud "
                    "already in device memory (cold_first also pays the handle's buffer allocations); cold_ms is timed around the ctypes "
                    "calls of this process, cold_c_caller_ms by tools/cold_path (the same three calls from C, a child process, the cloud "
                    "copied to the device right before it is timed; null when that binary was not built); new_clouds_through_queue_ms_by_lanes: a stream "
                    "of never-seen clouds through ppp_queue_* (handles taking turns), ms per cloud by number of lanes, same child process"}


def cold_path_from_c(args, engine, synth, base_seed):
    """The cold path without this process's ctypes layer: tools/cold_path (built by __graft_entry__.build() from tools/cold_path.cpp)
    on six never-seen clouds of the workload, written as binary PCDs to a temporary directory.  (min, median) in ms, or (None, None)."""
    import subprocess, tempfile
    exe = os.path.join(ROOT, "tools", "cold_path")
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if not os.path.exists(exe) or profiled:   # (no child process under a profiler's preloaded library)
        return None, None, None
    try:
        with tempfile.TemporaryDirectory() as d:
            names = []
            for k in range(6):
                p2, _ = synth.make_config(args.config, seed=base_seed + 15485863 * (k + 1))
                names.append(os.path.join(d, "c%d.pcd" % k))
                engine.save_pcd(names[-1], np.ascontiguousarray(p2), binary=True)
            r = subprocess.run([exe] + names, env=dict(os.environ, PPP_COLD_COPY_BEFORE="1", PPP_COLD_STREAM="1"), capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            sys.stderr.write("bench: tools/cold_path failed (%d): %s\n" % (r.returncode, (r.stdout + r.stderr)[-300:]))
            return None, None, None
        lines = r.stdout.strip().splitlines()
        last = [ln for ln in lines if ln.startswith("C caller")][-1]              # "... total min 88.8, median 89.7 us"
        a = last.split("total min ")[1]
        stream = {}
        for ln in lines:                                                          # "stream ... planner queue, 2 lane(s): 60 clouds in 2988.0 us = 49.8 us per cloud ..."
            if ln.startswith("stream of never-seen clouds") and " lane(s):" in ln:
                stream[int(ln.split("queue, ")[1].split(" lane")[0])] = float(ln.split(" = ")[1].split(" us per cloud")[0]) / 1e3
        return float(a.split(",")[0]) / 1e3, float(a.split("median ")[1].split(" ")[0]) / 1e3, (stream or None)
    except Exception as ex:
        sys.stderr.write("bench: tools/cold_path: %r\n" % (ex,))
        return None, None, None


def bench_slices(args, rank, local_rank, world, dev, dist, torch, engine, synth):
    """One cloud, slices sharded over the GPUs (no data-path collective until the gather); rank 0 finishes the list."""
    from polishpathplanning_amd.robot_path import exchange_counts, gather_robot_path, slice_ranges
    base_seed = sorted(synth.CONFIGS).index(args.config) + 1
    pts, cfg = synth.make_config(args.config, seed=base_seed)      # every rank generates the same cloud (a stand-in for the scan file)
    # The ranks agree on the whole cloud's bounds and point count with ONE all-reduce each of 3 minima, 3 maxima and a count over
    # the share of the file each of them read (here: a contiguous 1/N of the point array), in the planner's units (x 1000 in float)
    n_all = int(pts.shape[0])
    share = (pts[rank * n_all // world:(rank + 1) * n_all // world] * np.float32(1000)).astype(np.float32)
    fin = np.isfinite(share).all(axis=1)   # the engine's own count: ingest turns a point with ANY non-finite coordinate into NaN NaN NaN and bounds / counts skip those
    t_mn = torch.from_numpy(share[fin].min(axis=0) if fin.any() else np.full(3, np.inf, np.float32)).to(dev)
    t_mx = torch.from_numpy(share[fin].max(axis=0) if fin.any() else np.full(3, -np.inf, np.float32)).to(dev)
    t_n = torch.tensor([int(fin.sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(t_mn, op=dist.ReduceOp.MIN); dist.all_reduce(t_mx, op=dist.ReduceOp.MAX); dist.all_reduce(t_n, op=dist.ReduceOp.SUM)
    g_mn, g_mx, g_n = t_mn.cpu().numpy(), t_mx.cpu().numpy(), int(t_n.item())
    probe = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
    S = probe.range_interval(g_mn[0], g_mx[0])[2]
    probe.close()
    b, e = slice_ranges(S, world)[rank]
    if b >= e and rank == 0:  # more GPUs than slices: rank 0 still needs a handle, it finishes the list
        b, e = 0, 1
    eng = None
    nkept = max(S - 2, 0)
    counts_local = np.zeros(nkept, np.int32)
    w_local = 0
    n_part = 0
    if b < e:
        # this rank's part: the points of its slices' x interval (the pre-partition a real pipeline does while loading), with
        # their cloud indices; the handle never sees the rest of the cloud
        eng = engine.Engine(local_rank, tool_radius=cfg["tool_radius"], slice_begin=b, slice_end=e)
        lo, hi, _ = eng.range_interval(g_mn[0], g_mx[0])
        sx = (pts[:, 0] * np.float32(1000)).astype(np.float32)
        keep = np.nonzero((sx >= lo) & (sx <= hi))[0]
        n_part = int(len(keep))
        eng.set_cloud_part(pts[keep], keep, g_mn, g_mx, g_n, lo, hi)
        if slice_ranges(S, world)[rank][0] < slice_ranges(S, world)[rank][1]:
            eng.gen_path()
            w_local = eng.get_path()
            counts_local = eng.waypoint_counts()
    cnt = torch.from_numpy(counts_local.astype(np.int64)).to(dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    counts_all = cnt.cpu().numpy().astype(np.int32)                  # fixed workload: exchanged once
    w_ranks = exchange_counts(w_local, dist, dev)
    W = int(counts_all.sum())
    send = torch.zeros((max(w_local, 1), 6), dtype=torch.float32, device=dev)
    planner_stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev) if eng is not None else None

    def step():
        w = 0
        if eng is not None and w_local > 0:
            eng.run_async()                                          # a2..a12 of this rank's slices, one hipGraph launch
            w = eng.copy_stage_to_device(engine.STAGE_WP_PRESMOOTH, send.data_ptr(), send.shape[0])
        blocks = gather_robot_path(send[:w], dist, dev, w_ranks)
        if rank == 0:
            pre = torch.cat(blocks, dim=0).contiguous()
            # the gather and the cat run on the framework's stream, the planner on its own: order them on the GPU
            planner_stream.wait_stream(torch.cuda.current_stream())
            eng.finish_path_async(pre.data_ptr(), pre.shape[0], counts_all)   # a13..a15 once, over the whole list
            eng.sync()
            return pre
        return None

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
        assert got.shape[0] == W
        # untimed check of the last step: the list finished from the gathered blocks is the unsharded handle's list
        whole = engine.Engine(local_rank, tool_radius=cfg["tool_radius"])
        whole.set_cloud(pts); whole.gen_path(); whole.get_path()
        sharded_equals_whole = bool(np.array_equal(eng.waypoints(), whole.waypoints()))
        whole.close()
        n_points = int(pts.shape[0])
        alg_bytes = 12.0 * n_points + 24.0 * W
        out = {
            "metric": "polishing waypoints/sec for 1M-pt cloud, 256 slices; path L2 err vs ref",
            "value": W * args.steps / elapsed, "unit": "waypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "points_per_workpiece": n_points, "slices": S, "waypoints_per_workpiece": W,
                       "workpieces": 1, "parallelism": "slice ranges of one cloud, one range per GPU, every GPU holding only its own part "
                       "of the cloud (bounds and count agreed by all-reduce); RCCL gather of the pre-smoothing blocks; "
                       "postion_smooth/reduceRPY/flange once on rank 0", "points_on_rank0": n_part, "tool_radius_mm": cfg["tool_radius"]},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": alg_bytes / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole pipeline; every rank streams and indexes its own x interval of the cloud only"},
            "cpu_baseline": None,
            "assembled_path": {"sharded_list_equals_unsharded_handle": sharded_equals_whole, "rows": W},
        }
        emit_record(out)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not out["assembled_path"]["sharded_list_equals_unsharded_handle"]:
        raise SystemExit("bench.py --mode slices: the list assembled from the slice ranges differs from the unsharded handle's")
    return out


if __name__ == "__main__":
    main()
