"""Seeded synthetic workpiece clouds for the BASELINE.json configurations.

The reference ships no sample .pcd (its /PCD directory is git-ignored), so every
benchmark and parity test runs on clouds generated here.  Units are METRES, as in
the reference's PCD files (the planner multiplies by 1000 when ChangeRange=true,
src/Path_Alg/path_slicing_alg.cpp:19-23).  Layout: jittered 1.5 mm grid
(+-0.4 mm in x and y, so exact distance ties and x == plane have measure zero),
z = f(x, y) + 1500 mm (the sensor sits at the PCD VIEWPOINT, the origin, and looks
down +z: every normal flips consistently towards it), randomly permuted because
the reference's results depend on index order.
"""
import numpy as np

SPACING_MM = 1.5
JITTER_MM = 0.4
Z0_MM = 1500.0


def _surface(kind, x, y, amp):
    if kind == "dome":
        half = 0.5 * (x.max() - x.min())
        xc = x - 0.5 * (x.max() + x.min())
        return amp * (1.0 - (xc / half) ** 2)
    if kind == "wavy":
        return amp * np.sin(x / 600.0) * np.cos(y / 300.0)
    if kind == "blade":
        return amp * np.sin(y / 120.0) + 0.002 * x * y / 100.0
    if kind == "flat":
        return np.zeros_like(x)
    raise ValueError(kind)


def make_plate(nx, ny, kind="wavy", amp=40.0, seed=0, x0_mm=0.5, x_hi_mm=None, permute=True, z0_mm=Z0_MM):
    """nx x ny jittered grid; x from x0 (mm) upwards, y centred on 0.  Returns float32 [N,3] in metres.

    With x_hi_mm the x range is mapped affinely onto exactly [x0_mm, x_hi_mm], which pins the
    integer bounds the reference's slice walk truncates to (path_dynamic_alg.cpp:357-358).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    gx = (np.arange(nx, dtype=np.float64) * SPACING_MM + x0_mm)[:, None]
    gy = ((np.arange(ny, dtype=np.float64) - 0.5 * (ny - 1)) * SPACING_MM)[None, :]
    x = gx + rng.uniform(-JITTER_MM, JITTER_MM, (nx, ny))
    y = gy + rng.uniform(-JITTER_MM, JITTER_MM, (nx, ny))
    if x_hi_mm is not None:
        x = x0_mm + (x - x.min()) * ((x_hi_mm - x0_mm) / (x.max() - x.min()))
    z = _surface(kind, x, y, amp) + z0_mm
    pts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    if permute:
        pts = pts[rng.permutation(pts.shape[0])]
    return (pts / 1000.0).astype(np.float32)


# name -> (nx, ny, kind, amp, tool radius, description).  nx is chosen so that the
# centre-out integer walk of `connect` (path_dynamic_alg.cpp:308-372) yields the slice
# count BASELINE.json names; bench.py and the tests assert S from the generated cloud.
CONFIGS = {
    "cfg1_50k_s32": dict(nx=641, ny=78, x_hi_mm=961.5, kind="dome", amp=20.0, tool_radius=15.0, slices=32),
    "cfg2_1m_s256": dict(nx=2049, ny=488, x_hi_mm=3073.5, kind="wavy", amp=40.0, tool_radius=6.0, slices=256),
    "cfg3_250k_s128": dict(nx=1025, ny=244, x_hi_mm=1537.5, kind="dome", amp=25.0, tool_radius=6.0, slices=128),
    "cfg4_2m_s256": dict(nx=2049, ny=977, x_hi_mm=3073.5, kind="wavy", amp=40.0, tool_radius=6.0, slices=256),
    "cfg5_10m_s1024": dict(nx=8193, ny=1221, x_hi_mm=12289.5, kind="blade", amp=60.0, tool_radius=6.0, slices=1024),
    # small cases for the CPU test-suite and smoke()
    "tiny_5k": dict(nx=100, ny=50, kind="wavy", amp=8.0, tool_radius=6.0, slices=None),
    "small_40k": dict(nx=400, ny=100, kind="wavy", amp=20.0, tool_radius=6.0, slices=None),
}


def make_config(name, seed=None, **over):
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    if seed is None:
        seed = sorted(CONFIGS).index(name) + 1
    pts = make_plate(cfg["nx"], cfg["ny"], cfg["kind"], cfg["amp"], seed=seed, x_hi_mm=cfg.get("x_hi_mm"))
    return pts, cfg
