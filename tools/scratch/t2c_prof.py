import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from polishpathplanning_amd import engine, synth
pts, cfg = synth.make_config("cfg2_1m_s256")
for rep in range(3):
    e = engine.Engine(0, tool_radius=6.0); e.set_cloud(pts); e.nearest(pts[:1]*1000)
    t0=time.perf_counter(); e.trans2center(); print("t2c ms", (time.perf_counter()-t0)*1e3)
