"""Assembly of the final robot_path from per-GPU waypoint blocks.

Multi-GPU model: workpieces are independent, so a batch shards one
workpiece per GPU with no data-path collective.  The only exchange is the variable-length
gather of the finished W_g x 6 float blocks to rank 0 (the robot controller reads one file).
Blocks are ~0.6 MB per workpiece: latency-bound on xGMI, so one count exchange (once per batch) plus one
gather of padded blocks to rank 0 is used -- torch.distributed.gather, which RCCL runs as direct
send/recv pairs into the root, point to point over xGMI; no ring and no reduction exist anywhere on the path.
torch.distributed is plumbing here: backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import numpy as np


def exchange_counts(w_local, dist, device):
    """Waypoint count of every rank (one tiny all-gather).  A caller that replans the same batch
    can keep the result and pass it to gather_robot_path(counts=...)."""
    import torch
    cnt = torch.tensor([int(w_local)], dtype=torch.int64, device=device)
    counts = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, cnt)
    return counts.cpu().tolist()


def gather_robot_path(local_wp, dist=None, device=None, counts=None):
    """local_wp: torch tensor [W_local, 6] float32 on `device`.  Returns on rank 0 the list of
    per-rank waypoint tensors in rank order (None elsewhere).  Works for world_size 1 without
    a process group.  counts: per-rank waypoint counts if already known (skips the count exchange)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_wp]
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = local_wp.device if device is None else device
    if counts is None:
        counts = exchange_counts(local_wp.shape[0], dist, dev)
    elif counts[rank] != local_wp.shape[0]:
        raise ValueError("stale counts: rank %d has %d waypoints, not %d" % (rank, local_wp.shape[0], counts[rank]))
    wmax = max(max(counts), 1)
    send = torch.zeros((wmax, 6), dtype=torch.float32, device=dev)
    send[: local_wp.shape[0]] = local_wp
    if rank != 0:
        dist.gather(send, None, dst=0)
        return None
    recv = torch.empty((world * wmax, 6), dtype=torch.float32, device=dev)
    dist.gather(send, [recv[r * wmax: (r + 1) * wmax] for r in range(world)], dst=0)
    return [recv[r * wmax: r * wmax + counts[r]] for r in range(world)]


class RobotPathGatherer:
    """The same exchange with its buffers allocated once: the planner writes its W_local x 6 list straight into
    `send` (ppp_run_batch_async / ppp_copy_waypoints_to_device), gather() is then ONE collective and no copy.
    counts are exchanged at construction (a fixed batch has fixed counts; set_local_count() re-exchanges)."""

    def __init__(self, w_local, dist=None, device=None, force_collective=False):
        import torch
        # force_collective: run the collective even in a one-rank group (rehearsal of the exchange on a single GPU)
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)) else None
        self.device = device
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.torch = torch
        self.send = None
        self.set_local_count(w_local)

    def set_local_count(self, w_local):
        torch = self.torch
        self.w_local = int(w_local)
        self.counts = exchange_counts(self.w_local, self.dist, self.device) if self.dist else [self.w_local]
        wmax = max(max(self.counts), 1)
        if self.send is None or self.send.shape[0] != wmax:
            self.send = torch.zeros((wmax, 6), dtype=torch.float32, device=self.device)
            # only the root receives: a true gather (every other rank just sends its block)
            self.recv = torch.empty((self.world * wmax, 6), dtype=torch.float32, device=self.device) if (self.dist and self.rank == 0) else None
            self.recv_list = [self.recv[r * wmax: (r + 1) * wmax] for r in range(self.world)] if self.recv is not None else None
        self.wmax = wmax

    def gather_async(self, stream=None):
        """Enqueue the collective behind `stream`'s work without making that stream wait for it; returns the work handle
        (None without a process group).  blocks() is valid once the handle's wait() has been enqueued / has returned."""
        if not self.dist:
            return None
        if stream is not None:
            with self.torch.cuda.stream(stream):
                return self.dist.gather(self.send, self.recv_list, dst=0, async_op=True)
        return self.dist.gather(self.send, self.recv_list, dst=0, async_op=True)

    def blocks(self):
        """rank 0: list of per-rank [W_r, 6] views of the receive buffer in rank order; other ranks: None"""
        if not self.dist:
            return [self.send[: self.w_local]]
        if self.rank != 0:
            return None
        return [self.recv[r * self.wmax: r * self.wmax + self.counts[r]] for r in range(self.world)]

    def gather(self, stream=None):
        """rank 0: list of per-rank [W_r, 6] views in rank order; other ranks: None.
        stream: enqueue the collective behind this stream's work (default: the framework's current stream)"""
        if not self.dist:
            return [self.send[: self.w_local]]
        if stream is not None:
            with self.torch.cuda.stream(stream):
                self.dist.gather(self.send, self.recv_list, dst=0)
        else:
            self.dist.gather(self.send, self.recv_list, dst=0)
        if self.rank != 0:
            return None
        return [self.recv[r * self.wmax: r * self.wmax + self.counts[r]] for r in range(self.world)]


def run_pipelined_steps(count, plan, wait_planned, gatherers, wait_gathered, on_blocks=None):
    """`count` plan+gather steps with the gather of step k-1 overlapping the planning of step k.

    plan(k)          enqueues step k's planning; its lists land in gatherers[k % 2].send   (asynchronous)
    wait_planned()   host waits until the last plan() has finished
    wait_gathered()  host waits until the collectives enqueued so far have finished (their send buffer is free again)
    on_blocks(k, b)  optional: called on every rank with step k's gathered blocks (None off rank 0)

    Two gatherers = two send/receive buffer pairs, so step k+1 may overwrite the buffer of step k-1 only after
    wait_gathered().  Every step is planned AND gathered before this returns.  Returns the last step's blocks."""
    assert len(gatherers) == 2
    blocks = None
    for k in range(count):
        plan(k)
        if k > 0:
            blocks = gatherers[(k - 1) % 2].gather()
            if on_blocks:
                on_blocks(k - 1, blocks)
        wait_planned()
        wait_gathered()
    if count > 0:
        blocks = gatherers[(count - 1) % 2].gather()
        if on_blocks:
            on_blocks(count - 1, blocks)
        wait_gathered()
    return blocks


class StreamOrder:
    """Orders the planner's stream and the framework's (collective) stream with events; the host never waits for the
    step it has just enqueued.  Buffer b = step % 2.  planner_stream: torch.cuda.ExternalStream around Engine.stream_ptr()."""

    def __init__(self, torch, planner_stream):
        self.torch = torch
        self.ps = planner_stream
        self.planned = [torch.cuda.Event(), torch.cuda.Event()]
        self.gathered = [torch.cuda.Event(), torch.cuda.Event()]
        self.used = [False, False]

    def before_plan(self, b):
        # The gather that last read send[b] (two steps ago) must be over before the planner overwrites it.  The HOST
        # waits for that event -- it is two steps old, so the wait normally returns at once and merely bounds how far the
        # host runs ahead -- instead of the planner's stream: a cross-queue wait in front of every graph launch costs
        # the planner ~10 us per step, a host-side check of an old event costs it nothing.
        if self.used[b]:
            self.gathered[b].synchronize()

    def after_plan(self, b):
        self.planned[b].record(self.ps)

    def before_gather(self, b):
        self.torch.cuda.current_stream().wait_event(self.planned[b])

    def after_gather(self, b):
        self.gathered[b].record(self.torch.cuda.current_stream())
        self.used[b] = True


class NoOrder:
    """Synchronous back ends (the gloo tests): every call has finished when it returns."""

    def before_plan(self, b): pass
    def after_plan(self, b): pass
    def before_gather(self, b): pass
    def after_gather(self, b): pass


def run_streamed_steps(count, plan, gatherers, order, on_blocks=None):
    """`count` plan+gather steps with no host wait inside: step k's planning is enqueued, then step k-1's gather, and
    the two streams are ordered by `order` (StreamOrder: events).  The host runs ahead of the device; the caller
    synchronises once at the end (Engine.sync_batch + the framework stream).  Every step is planned and gathered.

    plan(k)           enqueues step k's planning into gatherers[k % 2].send
    on_blocks(k, b)   optional: enqueue the consumer of step k's gathered blocks (None off rank 0) on the framework stream
    Returns the last step's blocks."""
    assert len(gatherers) == 2
    blocks = None

    def gather(k):
        b = k % 2
        order.before_gather(b)
        out = gatherers[b].gather()
        if on_blocks:
            on_blocks(k, out)
        order.after_gather(b)
        return out

    for k in range(count):
        b = k % 2
        order.before_plan(b)
        plan(k)
        order.after_plan(b)
        if k > 0:
            blocks = gather(k - 1)
    if count > 0:
        blocks = gather(count - 1)
    return blocks


def run_chained_steps(count, plan, gatherers, planner_stream=None, on_blocks=None):
    """`count` plan+gather steps: step k's collective is enqueued behind step k's planning in the planner's own stream
    (asynchronously: the planner does not wait for it), and the planner waits for it only when it plans into the same buffer
    pair again, len(gatherers) steps later -- by then it has long finished.  No second framework stream and no host wait inside
    the loop; the collective of step k overlaps the planning of the steps behind it.
    (Measured on one GPU with a one-rank RCCL group, cfg 2: 0.166 ms per step, against 0.177 ms with the collective on the
    framework's default stream and events in between -- a third active hardware queue slows every dispatch -- and 0.156 ms
    with the planner waiting for each collective at once, which is the better choice only while the collective is as
    short as a one-rank copy.  Without any exchange a step takes 0.144 ms.)

    plan(k)           enqueues step k's planning into gatherers[k % H].send on the planner stream of step k
    gatherers         H >= 2 buffer pairs, used in turn
    planner_stream    torch.cuda.ExternalStream around Engine.stream_ptr() -- or a list of H of them when the steps take turns on
                      H engine handles (step k plans on handle k % H: its stream, its buffer pair; the framework runs the
                      collectives themselves one after the other on the communicator's own stream, in the order they are
                      called here, which is the same on every rank); None for synchronous back ends (gloo tests)
    on_blocks(k, b)   optional: called with step k's blocks once its collective has been waited for (None off rank 0)
    The caller synchronises the planner stream(s) once at the end.  Returns the last step's blocks."""
    import contextlib
    H = len(gatherers)
    assert H >= 2
    torch = gatherers[0].torch
    streams = planner_stream if isinstance(planner_stream, (list, tuple)) else [planner_stream] * H
    assert len(streams) == H

    def ctx(b):
        return torch.cuda.stream(streams[b]) if streams[b] is not None else contextlib.nullcontext()
    works = [None] * H
    pending = [None] * H        # step number whose collective is in flight on buffer pair b
    blocks = None

    def finish(b):
        nonlocal blocks
        if pending[b] is None:
            return
        if works[b] is not None:
            with ctx(b):
                works[b].wait()              # a stream wait (the host goes on), or a host wait for synchronous back ends
        blocks = gatherers[b].blocks()
        if on_blocks:
            on_blocks(pending[b], blocks)
        works[b] = None; pending[b] = None

    for k in range(count):
        b = k % H
        finish(b)
        plan(k)
        works[b] = gatherers[b].gather_async(streams[b])
        pending[b] = k
    for k in range(max(0, count - H), count):    # the last H steps, in order
        finish(k % H)
    return blocks


def slice_ranges(num_slices, world):
    """SURVEY.md 8e case (ii): GPU g plans the slices [g*S/world, (g+1)*S/world) of ONE cloud."""
    return [(g * num_slices // world, (g + 1) * num_slices // world) for g in range(world)]


def gather_slice_blocks(local_pre, local_counts, dist=None, device=None):
    """Slice-range sharding: every rank holds the pre-smoothing waypoints (a12 output, [W_g, 6] float32 on
    `device`) of its own slice range and the per-kept-slice counts (zero outside its range).  Returns on
    rank 0 (pre_all [W, 6] in slice order, counts_all int32[nkept]); (None, None) elsewhere.  postion_smooth
    couples neighbouring slices (path_translation_alg.cpp:117-140), so rank 0 then finishes the list once
    (Engine.finish_path_async).  Ranges ascend with the rank, so rank order is slice order."""
    import torch
    local_counts = np.ascontiguousarray(local_counts, np.int32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_pre, local_counts
    dev = local_pre.device if device is None else device
    cnt = torch.from_numpy(local_counts.astype(np.int64)).to(dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)  # disjoint supports: the sum is the global per-slice count vector
    blocks = gather_robot_path(local_pre, dist, dev)
    if dist.get_rank() != 0:
        return None, None
    return torch.cat(blocks, dim=0), cnt.cpu().numpy().astype(np.int32)


def concat_robot_path(blocks):
    """Rank-0 side: one W_total x 6 array, workpiece after workpiece."""
    import torch
    return torch.cat(list(blocks), dim=0) if len(blocks) else torch.zeros((0, 6))


def write_path_file(path, wp6):
    """The reference's pathFile format (path_translation_alg.cpp:216-228): six values per line,
    each followed by one blank, default ostream precision (6 significant digits)."""
    wp6 = np.asarray(wp6, dtype=np.float32)
    with open(path, "w") as f:
        for row in wp6:
            f.write("".join(_ostream_float(v) + " " for v in row) + "\n")


def _ostream_float(v):
    # std::ostream << float with default flags == printf("%g")
    return "%g" % float(v)


class RcclComm:
    """An ncclComm_t of this process's own, made through the librccl the engine itself resolves (dlopen("librccl.so"): the same
    library instance ppp_gather_waypoints calls into) -- for C++-style callers of ppp_gather_waypoints that own their communicator,
    and for its one-rank pre-flight.  world == 1 needs no rendezvous; world > 1 takes the 128-byte unique id from `broadcast`
    (a callable bytes -> bytes that hands rank 0's id to every rank, e.g. over torch.distributed)."""

    def __init__(self, rank=0, world=1, broadcast=None, lib_path=None):
        import ctypes as C
        import os
        name = lib_path or os.environ.get("PPP_RCCL_LIB") or "librccl.so"
        try:
            self.lib = C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            self.lib = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        self.lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = UniqueId()
        if rank == 0:
            rc = self.lib.ncclGetUniqueId(C.byref(uid))
            if rc:
                raise RuntimeError("ncclGetUniqueId failed (ncclResult %d)" % rc)
        if world > 1:
            raw = broadcast(bytes(uid.internal) if rank == 0 else b"\0" * 128)
            C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), int(world), uid, int(rank))
        if rc:
            raise RuntimeError("ncclCommInitRank failed (ncclResult %d)" % rc)
        self.ptr = self.comm.value

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
