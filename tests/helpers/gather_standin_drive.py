"""TEST INFRASTRUCTURE: runs ppp_gather_waypoints' multi-rank branch on one GPU against the recording RCCL stand-in
(tests/helpers/rccl_standin.c, loaded through PPP_RCCL_LIB).  Prints one JSON line.  Started as its own process by
tests/test_gpu_parity.py so that the engine's one-time librccl lookup sees the stand-in."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    work = sys.argv[1]
    so = os.path.join(work, "librccl_standin.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tests", "helpers", "rccl_standin.c")])
    log = os.path.join(work, "rccl.log")
    os.environ["PPP_RCCL_LIB"] = so
    os.environ["PPP_RCCL_STANDIN_LOG"] = log
    import numpy as np
    from polishpathplanning_amd import engine, synth
    from polishpathplanning_amd.hipbuf import DeviceBuffer
    pts, cfg = synth.make_config("small_40k")
    e = engine.Engine(0, tool_radius=6.0)
    e.set_cloud(pts); e.gen_path(); W = e.get_path()
    own = e.waypoints()
    counts = [7, 0, W, 11]            # this process plays rank 2 = the root of a 4-rank communicator
    total = sum(counts)
    buf = DeviceBuffer(total * 24)
    out = {"W": int(W), "counts": counts, "recv_ptr": int(buf.ptr)}

    def read_log():
        lines = open(log).read().strip().split("\n") if os.path.exists(log) else []
        if os.path.exists(log):
            os.remove(log)
        return lines

    e.gather_waypoints(0x1234, 2, 4, 2, counts, buf.ptr)
    e.sync()
    got = buf.to_host(total * 6).reshape(total, 6)
    out["root_log"] = read_log()
    out["own_block_in_place"] = bool(np.array_equal(got[7:7 + W], own))
    out["stream"] = int(e.stream_ptr())
    # a sending rank (rank 3 of 4, root 2): one send of its list
    e.gather_waypoints(0x1234, 3, 4, 2, [7, 0, 5, W], 0)
    out["send_log"] = read_log()
    # a failing receive: the group is closed all the same and the error comes back
    os.environ["PPP_RCCL_STANDIN_FAIL"] = "recv"
    try:
        e.gather_waypoints(0x1234, 2, 4, 2, counts, buf.ptr)
        out["fail_raised"] = False
    except engine.PPPError as ex:
        out["fail_raised"] = True
        out["fail_msg"] = str(ex)
    out["fail_log"] = read_log()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
