"""A caller-owned device buffer without torch: hipMalloc / hipMemcpy / hipFree through ctypes, on the HIP runtime the
engine library already loaded.  Used where a process must not pull torch's own ROCm runtime in (tests, tools); a
framework tensor's data_ptr() serves the same purpose in bench.py."""
import ctypes as C

import numpy as np

_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    return _hip


class DeviceBuffer:
    def __init__(self, nbytes):
        self.hip = _rt()
        p = C.c_void_p()
        rc = _rt().hipMalloc(C.byref(p), C.c_size_t(max(int(nbytes), 1)))
        if rc:
            raise MemoryError("hipMalloc(%d) failed: %d" % (nbytes, rc))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def upload(self, arr):
        """Synchronous copy of a C-contiguous array into the buffer."""
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        rc = _rt().hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1)
        if rc:
            raise RuntimeError("hipMemcpy H2D failed: %d" % rc)

    def to_host(self, nfloats):
        out = np.empty(int(nfloats), np.float32)
        if nfloats:
            rc = _rt().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(4 * int(nfloats)), 2)
            if rc:
                raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
        return out

    def free(self):
        if self.ptr:
            _rt().hipFree(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
