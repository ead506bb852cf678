#!/usr/bin/env python3
"""What a same-queue exchange would cost a step: the planner's graph launch followed by a device-to-device
copy of the finished list on the SAME stream, against the graph launch alone.  (bench.py's N > 1 loop runs the collective
on RCCL's own stream; its one-rank rehearsal measures 0.161 ms per step.)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from polishpathplanning_amd import engine, hipbuf, synth  # noqa: E402

pts, cfg = synth.make_config("cfg2_1m_s256")
e = engine.Engine(0, tool_radius=6.0)
e.set_cloud(pts); e.gen_path(); W = e.get_path()
send = hipbuf.DeviceBuffer(W * 24 * 2)
recv = hipbuf.DeviceBuffer(W * 24 * 2)
hip = C.CDLL("libamdhip64.so")
stream = C.c_void_p(e.stream_ptr())
offs = np.zeros(1, np.int64)


def timeit(f, n=300):
    f(0); f(1); e.sync()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for k in range(n):
            f(k)
        e.sync()
        best = min(best, (time.perf_counter() - t) / n)
    return best * 1e3


def plan(k):
    engine.run_batch_async([e], send.ptr + (k % 2) * W * 24, offs, [W])


def plan_copy(k):
    plan(k)
    hip.hipMemcpyAsync(C.c_void_p(recv.ptr + (k % 2) * W * 24), C.c_void_p(send.ptr + (k % 2) * W * 24), C.c_size_t(W * 24), C.c_int(3), stream)


print("plan only                      %.4f ms per step" % timeit(plan))
print("plan + same-stream 0.6 MB copy %.4f ms per step" % timeit(plan_copy))
