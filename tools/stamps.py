"""Developer tool: per-phase s_memtime shares of the instrumented kernels (diagnostic build,
`make -C polishpathplanning_amd/csrc stamps`).  Read the SHARES, not the absolute times: the
stamps serialise what the product build overlaps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polishpathplanning_amd import engine, synth
argv = sys.argv[1:]
stamps_lib = "libppp_hip_stamps.so"
if "--lib" in argv:
    stamps_lib = argv[argv.index("--lib") + 1]
    del argv[argv.index("--lib"):argv.index("--lib") + 2]
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), stamps_lib)
args = [a for a in argv if not a.startswith("--")]
name = args[0] if args else "cfg2_1m_s256"
dynamic = 1 if "--dynamic" in sys.argv else 0
pts, cfg = synth.make_config(name)
e = engine.Engine(0, tool_radius=cfg["tool_radius"], dynamic_adjustment=dynamic); e.set_cloud(pts)
out = (C.c_ulonglong * 256)()
e.gen_path(); e.get_path(); engine.lib().ppp_dbg_stamps(out)  # warm-up + reset
N = 10
for _ in range(N):
    e.gen_path_async(); e.get_path_async()
e.sync(); engine.lib().ppp_dbg_stamps(out)
labels = {0: ("k_slice_kd", ["gather", "band sort", "NN+lerp", "cand sort", "flatten"]),
          1: ("k_pose", ["staging", "dy+spline", "nearest", "normal", "pose+handeye"]),
          2: ("k_slab_scatter", ["zero", "count", "reserve", "scatter"]),
          5: ("k_setup", ["partials", "slab scan", "bounds+walk", "band limits"]),
          7: ("k_win_scatter", ["loads issued + table", "points in, windows, LDS ranks", "global atomics", "stores issued", "bounds partial"]),
          6: ("k_win_slice", ["load+clear", "hist+scan+place", "bucket finish", "pairing NN", "cand sort", "flatten", "knots+spline", "nearest", "normal", "pose+store"]),
          }
if dynamic:
    labels[3] = ("k_dyn_boundary_pts", ["prologue+dy", "spline", "knn search", "knn rank", "eigen", "extremum", "store", "-", "-",
                                         "normals gather", "rank-order sums", "axes", "ellipse points"])
    labels[4] = ("k_dyn_adjust_pts", ["prologue", "spline", "knn search (all)", "knn rank (all)", "eigen", "extremum",
                                      "boundary eval", "-", "snap store", "normals gather", "rank-order sums", "axes", "ellipse points",
                                      "first reads", "fit"])
for kid, (kn, ls) in labels.items():
    vals = [out[16 * kid + i] / N for i in range(len(ls))]
    tot = sum(vals) or 1
    print("%-16s total %8.0f ticks (%.1f us): " % (kn, tot, tot / 2400.0) + "  ".join("%s %.0f%%" % (l, 100 * v / tot) for l, v in zip(ls, vals)))
