"""Device buffers for the GPU tests through the HIP runtime the library itself is linked against (ctypes), so that a
test needs no second GPU runtime in the process."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipStreamDestroy.argtypes = [C.c_void_p]
        _hip.hipStreamQuery.argtypes = [C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


def mem_free():
    """Free device memory in bytes (hipMemGetInfo) of the current device."""
    free, total = C.c_size_t(), C.c_size_t()
    check(hip().hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value


def check(status, what):
    if status != 0:
        raise RuntimeError("%s failed with HIP status %d" % (what, status))


class DeviceArray:
    """n float32 values on the current device."""

    def __init__(self, n, fill_byte=0xff):
        self.n = int(n)
        self.ptr = C.c_void_p()
        check(hip().hipMalloc(C.byref(self.ptr), self.n * 4), "hipMalloc")
        check(hip().hipMemset(self.ptr, fill_byte, self.n * 4), "hipMemset")

    def numpy(self):
        check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        out = np.empty(self.n, dtype=np.float32)
        check(hip().hipMemcpy(out.ctypes.data, self.ptr, self.n * 4, 2), "hipMemcpy")  # 2 = device to host
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    __del__ = free


class Stream:
    def __init__(self):
        self.handle = C.c_void_p()
        check(hip().hipStreamCreate(C.byref(self.handle)), "hipStreamCreate")

    def done(self):
        """True when everything enqueued on the stream has finished (hipStreamQuery == hipSuccess)."""
        return hip().hipStreamQuery(self.handle) == 0

    def synchronize(self):
        check(hip().hipStreamSynchronize(self.handle), "hipStreamSynchronize")

    def close(self):
        if self.handle:
            hip().hipStreamDestroy(self.handle)
            self.handle = C.c_void_p()

    __del__ = close
