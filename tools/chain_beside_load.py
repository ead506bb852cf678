"""Developer tool: the dynamic-adjustment pass (cfg 2) alone and while a second handle keeps the device full with plain passes
(what concurrent launches cost the chain).  usage: python tools/chain_beside_load.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth
pts, cfg = synth.make_config("cfg2_1m_s256")
A = engine.Engine(0, tool_radius=cfg["tool_radius"], dynamic_adjustment=1); A.set_cloud(pts)
A.run_async(); A.sync()
def timed():
    t = time.perf_counter(); A.run_async(); A.sync(); return (time.perf_counter() - t) * 1e3
print("alone: " + " ".join("%.3f" % timed() for _ in range(5)))
for name in ("cfg5_10m_s1024", "cfg2_1m_s256"):
    p2, c2 = synth.make_config(name)
    B = engine.Engine(0, tool_radius=c2["tool_radius"]); B.set_cloud(p2); B.run_async(); B.sync()
    t = time.perf_counter()
    for _ in range(20): B.run_async()
    B.sync(); tb = (time.perf_counter() - t) / 20 * 1e3
    res = []
    for _ in range(5):
        for _ in range(int(12 / tb) + 2): B.run_async()      # ~12 ms of background passes
        time.sleep(0.0005)
        res.append(timed())
        B.sync()
    print("beside %s passes (%.3f ms each): " % (name, tb) + " ".join("%.3f" % r for r in res))
