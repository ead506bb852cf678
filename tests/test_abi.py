"""The C-ABI library: it builds for gfx950 without a GPU, exports every symbol the header
declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ppp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ppp_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol(engine_mod):
    engine_mod.build()
    L = ctypes.CDLL(engine_mod.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(engine_mod.EXPORTS)


def test_library_is_gfx950_code_object(engine_mod):
    blob = open(engine_mod.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_slice" in blob and b"k_slab_scatter" in blob


def test_default_params_match_config_txt(engine_mod):
    p = engine_mod.default_params()
    # /root/reference/config.txt:1-13 and Path_Generate_Algorithm.h:43-48
    assert (p.tool_radius, p.path_resolution, p.rpy_resolution) == (12.0, 7.0, 7.0)
    assert abs(p.ee_length - 0.3) < 1e-7 and p.change_range == 1
    assert [round(v, 6) for v in p.handeye] == [-0.764091, 0.025886, 0.66379, -3.127017, -0.040124, -1.606358]


def test_no_device_fails_loudly(engine_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine_mod.PPPError) as ei:
        engine_mod.Engine(0)
    assert ei.value.code == engine_mod.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: the package, the C ABI headers
    and drop-in classes, the example CLIs and the developer tools under tools/ never do"""
    for sub in ("polishpathplanning_amd", "include", "examples", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".sh")):
                    src = open(os.path.join(dp, f), errors="ignore").read()
                    assert "ppp_oracle" not in src and "from oracle" not in src and "import oracle" not in src and "import ppo" not in src, f


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/ppp_hip.h must compile as C (what cgo / a JNI stub / ctypesgen would see)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "ppp_hip.h"\nint main(void) { ppp_params p; ppp_config c; ppp_default_params(&p); ppp_default_config(&c); return 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
