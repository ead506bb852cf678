"""Driven by tests/test_host_logic.py::test_pcd_and_config_parsers_under_address_sanitizer with libasan preloaded: the host
I/O of the C ABI (ppp_io.cpp, compiled alone with -fsanitize=address,undefined) on valid files of the three PCD encodings,
on thousands of mutated / truncated copies of them, and on mutated config files.  Nothing may crash or read out of bounds;
a file is either loaded or refused."""
import ctypes as C
import os
import sys

import numpy as np

lib = C.CDLL(sys.argv[1])
tmp = sys.argv[2]
fp = C.POINTER(C.c_float)
lib.ppp_load_pcd.argtypes = [C.c_char_p, C.POINTER(fp), C.POINTER(C.c_size_t), fp]
lib.ppp_save_pcd.argtypes = [C.c_char_p, fp, C.c_size_t, C.c_size_t, fp, C.c_int]
lib.ppp_free.argtypes = [C.c_void_p]


def load(path):
    p = fp(); n = C.c_size_t(); vp = (C.c_float * 7)()
    rc = lib.ppp_load_pcd(path.encode(), C.byref(p), C.byref(n), vp)
    if rc == 0:
        a = np.ctypeslib.as_array(p, shape=(max(n.value, 1) * 3,))[: n.value * 3].copy()
        lib.ppp_free(p)
        return a.reshape(-1, 3)
    return None


rng = np.random.default_rng(5)
pts = rng.normal(size=(700, 3)).astype(np.float32)
pts[3] = np.nan
vp = (C.c_float * 7)(0.1, 0.2, 0.3, 1, 0, 0, 0)
loaded = refused = 0
for mode in (0, 1, 2):
    good = os.path.join(tmp, "g%d.pcd" % mode)
    assert lib.ppp_save_pcd(good.encode(), pts.ctypes.data_as(fp), len(pts), 3, vp, mode) == 0
    back = load(good)
    assert back is not None and back.shape == pts.shape
    raw = open(good, "rb").read()
    bad = os.path.join(tmp, "b.pcd")
    for t in range(1500):
        b = bytearray(raw)
        kind = t % 4
        if kind == 0:                       # flip a few bytes anywhere
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:                     # truncate
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 2:                     # garble the header only
            hdr_end = raw.find(b"DATA")
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, max(hdr_end, 1) + 20))] = int(rng.integers(32, 127))
        else:                               # a huge count in the header
            b = bytearray(raw.replace(b"POINTS 700", b"POINTS %d" % int(rng.integers(10 ** 6, 2 ** 40))).replace(b"WIDTH 700", b"WIDTH 99999999999"))
        open(bad, "wb").write(bytes(b))
        r = load(bad)
        if r is None:
            refused += 1
        else:
            loaded += 1
            assert r.ndim == 2 and r.shape[1] == 3
# headers whose SIZE / COUNT would move the x, y, z offsets outside the record (negative, zero, huge, overflowing)
body = pts[:5].tobytes() + b"\0" * 64
for hdr in ("FIELDS a x y z\nSIZE -8 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n", "FIELDS a x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT -3 1 1 1\n",
            "FIELDS x y z\nSIZE 0 4 4\nTYPE F F F\nCOUNT 1 1 1\n", "FIELDS a x y z\nSIZE 8 4 4 4\nTYPE F F F F\nCOUNT 2147483647 1 1 1\n",
            "FIELDS x y z\nSIZE 3 4 4\nTYPE F F F\nCOUNT 1 1 1\n", "FIELDS a x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1073741824 1 1 1\n"):
    for kind in ("binary", "binary_compressed", "ascii"):
        payload = body if kind != "ascii" else b"1 2 3 4\n" * 5
        if kind == "binary_compressed":
            payload = (60).to_bytes(4, "little") + (60).to_bytes(4, "little") + body
        open(bad, "wb").write(("VERSION 0.7\n" + hdr + "WIDTH 5\nHEIGHT 1\nPOINTS 5\nDATA " + kind + "\n").encode() + payload)
        assert load(bad) is None, (hdr, kind)
        refused += 1
print("pcd: %d loaded, %d refused" % (loaded, refused))


class Params(C.Structure):
    _fields_ = [("raw", C.c_char * 4096)]


cfgbuf = (C.c_char * 8192)()                # ppp_config is smaller than this
lib.ppp_default_config.argtypes = [C.c_void_p]
lib.ppp_read_config.argtypes = [C.c_char_p, C.c_void_p]
text = ("Tool_Radius = 6\npathFile = out.txt\nPathResolution = 7\nRPYresolution = 7\nEnd effector length = 0.3\nSmooth = false\n"
        "Alignment = false\nChangeRange = true\nRemoveOutlier = false\nDynamic_adjustment = true\nAdjust_Threshold = 1\ntoolthickness = 10\ndepth = 0.01\n# comment\n")
cp = os.path.join(tmp, "c.txt")
for t in range(800):
    b = bytearray(text.encode())
    for _ in range(int(rng.integers(0, 8))):
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    if t % 5 == 0:
        b = b[: int(rng.integers(0, len(b)))]
    if t % 7 == 0:
        b += b"pathFile = " + b"x" * int(rng.integers(100, 5000)) + b"\n"
    open(cp, "wb").write(bytes(b))
    lib.ppp_default_config(cfgbuf)
    lib.ppp_read_config(cp.encode(), cfgbuf)
lib.ppp_read_config(os.path.join(tmp, "missing.txt").encode(), cfgbuf)
lib.ppp_write_path_file.argtypes = [C.c_char_p, fp, C.c_size_t]
w = rng.normal(size=(40, 6)).astype(np.float32)
assert lib.ppp_write_path_file(os.path.join(tmp, "w.txt").encode(), w.ctypes.data_as(fp), 40) == 0
assert lib.ppp_write_path_file(os.path.join(tmp, "nodir", "w.txt").encode(), w.ctypes.data_as(fp), 40) != 0
print("done")
