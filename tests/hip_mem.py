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
        _hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        _hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
        _hip.hipGraphDestroy.argtypes = [C.c_void_p]
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


class Graph:
    """What was enqueued on `stream` between begin() and end(), as an executable HIP graph."""

    def __init__(self, stream):
        self.stream, self.graph, self.exe = stream, C.c_void_p(), C.c_void_p()

    def begin(self):
        check(hip().hipStreamBeginCapture(self.stream.handle, 0), "hipStreamBeginCapture")  # 0 = global mode

    def end(self):
        check(hip().hipStreamEndCapture(self.stream.handle, C.byref(self.graph)), "hipStreamEndCapture")
        check(hip().hipGraphInstantiate(C.byref(self.exe), self.graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")

    def launch(self):
        check(hip().hipGraphLaunch(self.exe, self.stream.handle), "hipGraphLaunch")

    def close(self):
        if self.exe:
            hip().hipGraphExecDestroy(self.exe)
            self.exe = C.c_void_p()
        if self.graph:
            hip().hipGraphDestroy(self.graph)
            self.graph = C.c_void_p()

    __del__ = close


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
