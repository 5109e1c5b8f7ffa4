"""GPU: rt_render_device_noalloc — the render entry point that allocates nothing — against rt_render_device.

Called through the C ABI on device buffers the test allocates itself (tests/hip_mem.py).  Two calls in a
row, the second on a smaller input: the fixed scratch array keeps the first call's contents, so the second call
shows that what must start at zero is cleared and that stale partial sums are never read.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(fn, scene, cam, spp, depth, seed, samples_per_item, rt, stream=None):
    import hip_mem

    out = hip_mem.DeviceArray(cam.vert_res * cam.horiz_res * 3)  # filled with NaN bit patterns
    o = rt.Scene.opts(0, spp, depth, seed, 0, samples_per_item)
    status = fn(scene.handle, C.byref(cam.c), C.byref(o), out.ptr, stream)
    result = out.numpy().reshape(cam.vert_res, cam.horiz_res, 3)
    out.free()
    return status, result


def test_noalloc_render_equals_render_device_twice_in_a_row(rt):
    import hip_mem

    from raytracer_amd import _abi

    lib = _abi.lib
    scene = rt.Scene(rt.gen_scene(1))
    limit = _abi.RT_NOALLOC_SCRATCH_BYTES - 2048
    # (frame, spp, samples per item or 0): first a chunked render that fills a good part of the scratch,
    # then a smaller one whose slots lie inside the first one's
    for (w, h, spp, spi) in ((800, 450, 16, 4), (320, 200, 12, 0)):
        cam = rt.default_camera(w, h)
        st_a, a = _render(lib.rt_render_device_noalloc, scene, cam, spp, 50, 3, spi, rt)
        st_b, b = _render(lib.rt_render_device, scene, cam, spp, 50, 3, spi, rt)
        assert st_a == 0 and st_b == 0, lib.rt_last_error()
        chunk = scene.render_chunk(cam, spp, samples_per_item=spi)
        assert -(-spp // chunk) * w * h * 12 <= limit and -(-spp // chunk) > 1  # chunked, and it fits: same chunking on both sides
        assert np.isfinite(a).all() and a.min() >= 0
        assert np.array_equal(a, b), "noalloc render differs from rt_render_device at %dx%d" % (w, h)
    # chunk size left to the library on a frame whose usual chunking does not fit: fewer, longer chunks, by the stated rule
    cam = rt.default_camera(1280, 720)
    spp = 30
    fit = max(limit // (1280 * 720 * 12), 1)
    assert -(-spp // scene.render_chunk(cam, spp)) > fit  # the usual chunking would not fit
    st_a, a = _render(lib.rt_render_device_noalloc, scene, cam, spp, 50, 5, 0, rt)
    st_b, b = _render(lib.rt_render_device, scene, cam, spp, 50, 5, -(-spp // fit), rt)
    assert st_a == 0 and st_b == 0 and np.array_equal(a, b)
    # a chunking given by the caller that does not fit is refused, with the limit in the message
    st, _ = _render(lib.rt_render_device_noalloc, scene, cam, spp, 50, 5, 1, rt)
    assert st == _abi.RT_ERR_LIMIT and "scratch" in lib.rt_last_error().decode()
    # and the next call still works, on a non-default stream
    s2 = hip_mem.Stream()
    cam = rt.default_camera(200, 120)
    st_a, a = _render(lib.rt_render_device_noalloc, scene, cam, 6, 50, 8, 2, rt, stream=s2.handle)
    st_b, b = _render(lib.rt_render_device, scene, cam, 6, 50, 8, 2, rt)
    assert st_a == 0 and np.array_equal(a, b)


def test_noalloc_render_leaves_free_memory_alone_and_replays_from_a_graph(rt):
    """The two kinds of caller the entry point is for.  A serving loop: after one warm-up call (which asks the
    scratch array's address and loads the code object) further calls leave the device's free memory exactly as it was.
    Code under HIP graph capture: the call is captured on a stream of the caller's, and two replays of the graph
    give the image of the plain render."""
    import hip_mem

    from raytracer_amd import _abi

    lib = _abi.lib
    scene = rt.Scene(rt.gen_scene(1))
    cam = rt.default_camera(640, 360)
    spp, spi = 12, 4  # three chunks: partial sums and the combine kernel are part of the call
    o = rt.Scene.opts(0, spp, 50, 6, 0, spi)
    out = hip_mem.DeviceArray(640 * 360 * 3)
    stream = hip_mem.Stream()
    assert lib.rt_render_device_noalloc(scene.handle, C.byref(cam.c), C.byref(o), out.ptr, stream.handle) == 0, lib.rt_last_error()
    stream.synchronize()
    want = out.numpy().copy()
    free0 = hip_mem.mem_free()
    for _ in range(3):
        assert lib.rt_render_device_noalloc(scene.handle, C.byref(cam.c), C.byref(o), out.ptr, stream.handle) == 0
    stream.synchronize()
    assert hip_mem.mem_free() == free0
    st_b, plain = _render(lib.rt_render_device, scene, cam, spp, 50, 6, spi, rt)
    assert st_b == 0 and np.array_equal(want.reshape(360, 640, 3), plain)
    # capture one call, replay it twice into a buffer wiped in between
    g = hip_mem.Graph(stream)
    g.begin()
    status = lib.rt_render_device_noalloc(scene.handle, C.byref(cam.c), C.byref(o), out.ptr, stream.handle)
    g.end()
    assert status == 0, lib.rt_last_error()
    for _ in range(2):
        hip_mem.check(hip_mem.hip().hipMemset(out.ptr, 0xff, out.n * 4), "hipMemset")
        g.launch()
        stream.synchronize()
        assert np.array_equal(out.numpy(), want)
    g.close()
    # and the handle still renders the ordinary way afterwards
    st_c, again = _render(lib.rt_render_device, scene, cam, spp, 50, 6, spi, rt)
    assert st_c == 0 and np.array_equal(again, plain)
