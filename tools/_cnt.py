import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libppp_hip_cnt.so")
for name in sys.argv[1:]:
    pts, cfg = synth.make_config(name)
    for walk in (1, 2):
        e = engine.Engine(0, tool_radius=cfg["tool_radius"], dynamic_adjustment=1, walk=walk); e.set_cloud(pts)
        e.gen_path(); e.get_path()
        v = e.smooth_sweeps()
        print(name, "walk", walk, "nodes", v & 65535, "with >1 evaluation", v >> 16)
