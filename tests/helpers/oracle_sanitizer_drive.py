"""Driven by tests/test_oracle_crosschecks.py::test_oracle_under_address_sanitizer with libasan preloaded: every entry
point of the oracle once, on the -fsanitize=address,undefined build (oracle/Makefile `asan`)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import ppo
ppo._LIB_PATH = os.path.join(os.path.dirname(ppo.__file__), "libppp_oracle_asan.so")
from polishpathplanning_amd import synth
pts, cfg = synth.make_config("tiny_5k")
pts = pts.copy(); pts[7] = np.nan
for walk, pairing, dyn in [(1, 0, 0), (3, 1, 1), (2, 0, 1), (0, 0, 0)]:
    o = ppo.Oracle(pts, tool_radius=6.0, walk=walk, pairing=pairing, dynamic_adjustment=dyn)
    print("gen", o.gen_path(), "path", o.get_path())
o = ppo.Oracle(pts, tool_radius=6.0)
print("vox", o.voxel_down(0.1, 1, 1)); print("mls", o.smooth_mls(15.0, 3)); print("sor", o.remove_outlier(50, 1.0)[0])
a = 0.3
R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
q = (pts.astype(np.float64) @ R.T + 0.2).astype(np.float32)
o = ppo.Oracle(q, tool_radius=6.0)
print("t2c", o.trans2center()[0]); print("gen", o.gen_path(), "path", o.get_path())
print(ppo.eigensolver3f(np.diag([3., 2., 1.]).astype(np.float32))[0])
o = ppo.Oracle(pts[:50], tool_radius=6.0)
print("tiny", o.gen_path(), o.voxel_down(5, 5, 5), o.smooth_mls(15.0, 3), o.trans2center()[0])
print("done")
