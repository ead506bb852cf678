#!/usr/bin/env python3
"""Dynamic adjustment on one synthetic config: GPU time per kernel, and the knots /
waypoints against the oracle.  usage: python tests/tools/dyn_check.py [config] [walk] [--no-oracle]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polishpathplanning_amd import engine, synth  # noqa: E402


def main():
    if "--lib" in sys.argv:
        i = sys.argv.index("--lib")
        engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2_1m_s256"
    walk = int(args[1]) if len(args) > 1 else 1
    kw = dict(tool_radius=6.0, walk=walk, dynamic_adjustment=1)
    if walk == 3:
        kw.update(pairing=1, curvature_k=10, depth=0.005)
    pts, cfg = synth.make_config(name)
    e = engine.Engine(0, **kw)
    e.set_cloud(pts)
    S = e.gen_path()
    W = e.get_path()
    t = time.perf_counter()
    for _ in range(5):
        e.gen_path_async(); e.get_path_async(); e.sync()
    tg = (time.perf_counter() - t) / 5
    e.run_async(); e.sync()
    t = time.perf_counter()
    for _ in range(5):
        e.run_async(); e.sync()
    tgraph = (time.perf_counter() - t) / 5
    e.enable_timing(True)
    e.gen_path_async(); e.get_path_async(); e.sync()
    kt, kl = e.kernel_times(with_launches=True)
    e.enable_timing(False)
    print(name, "walk", walk, "S", S, "W", W, "gpu %.3f ms (plain launches), %.3f ms (hipGraph replay)" % (tg * 1e3, tgraph * 1e3))
    print({k: (round(v, 3), kl[k]) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]) if v > 0.02})
    if "--no-oracle" in sys.argv:
        return
    from oracle import ppo
    o = ppo.Oracle(pts, **kw)
    t = time.perf_counter(); So = o.gen_path(); Wo = o.get_path(); to = time.perf_counter() - t
    bad = sum(1 for s in range(So) if not all(np.array_equal(a, b) for a, b in zip(e.nodes(s), o.nodes(s))))
    d = np.linalg.norm(e.waypoints()[:, :3] - o.waypoints()[:, :3], axis=1) if W == Wo else np.array([-1.0])
    print("oracle %.2f s; W %d/%d; slices with different knots %d/%d; max err %.2e m" % (to, W, Wo, bad, So, d.max()))
    assert bad == 0 and W == Wo and d.max() <= 1e-4


if __name__ == "__main__":
    main()
