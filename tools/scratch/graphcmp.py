import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from polishpathplanning_amd import engine, synth, hipbuf
pts, cfg = synth.make_config("cfg2_1m_s256")
e = engine.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.gen_path(); W = e.get_path()
buf = hipbuf.DeviceBuffer(W * 24 * 2)
def timeit(f, n=200):
    f(0); f(1); e.sync()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for k in range(n): f(k)
        e.sync()
        best = min(best, (time.perf_counter() - t) / n)
    return best * 1e3
print("run_async               %.4f ms" % timeit(lambda k: e.run_async()))
print("run_batch_async no dst  %.4f ms" % timeit(lambda k: engine.run_batch_async([e])))
offs = np.zeros(1, np.int64)
print("run_batch_async dst     %.4f ms" % timeit(lambda k: engine.run_batch_async([e], buf.ptr + (k % 2) * W * 24, offs, [W])))
print("run_batch_async 1 dst   %.4f ms" % timeit(lambda k: engine.run_batch_async([e], buf.ptr, offs, [W])))
