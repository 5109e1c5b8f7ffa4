"""ctypes access to the CPU oracle (oracle/build/librt_oracle.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "build", "librt_oracle.so")
ORACLE_SELFTEST = os.path.join(ORACLE_DIR, "build", "selftest")

TRAVERSE_LINEAR, TRAVERSE_TREE, TRAVERSE_GATED_BRUTE = 0, 1, 2
SAMPLER_REFERENCE, SAMPLER_DIRECT = 0, 1

HITTABLE_DTYPE = np.dtype(
    [
        ("shape", "<i4"), ("material", "<i4"), ("centre", "<f8", (3,)), ("normal", "<f8", (3,)), ("radius", "<f8"),
        ("albedo", "<f8", (3,)), ("fuzz", "<f8"), ("refractive_index", "<f8"), ("orient_up", "<f8", (3,)),
        ("orient_around", "<f8", (3,)), ("texture", "<i4"), ("reserved", "<i4"),
    ],
    align=True,
)


class Camera(C.Structure):
    _fields_ = [
        ("lookat", C.c_double * 3), ("lookfrom", C.c_double * 3), ("lookup", C.c_double * 3),
        ("focal_distance", C.c_double), ("inv_focal_length", C.c_double), ("aperture", C.c_double),
        ("horiz_arm", C.c_double * 3), ("vert_arm", C.c_double * 3),
        ("horiz_res", C.c_uint32), ("vert_res", C.c_uint32), ("aspect_ratio", C.c_double),
    ]


class Texture(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


def build(force=False):
    """Compiles the oracle in place (g++, seconds).  No-op when it is already built."""
    if force or not (os.path.exists(ORACLE_LIB) and os.path.exists(ORACLE_SELFTEST)):
        subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_LIB)
        vp, sz, u32, u64, i, d = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_double
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, sz, C.POINTER(Texture), sz]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_build_tree.restype = C.c_long
        L.orc_scene_build_tree.argtypes = [vp]
        L.orc_scene_tree_depth.argtypes = [vp]
        L.orc_scene_set_tie_rule.argtypes = [vp, i]
        L.orc_scene_set_dir_noise.argtypes = [vp, d]
        L.orc_camera_build.argtypes = [C.POINTER(d), C.POINTER(d), d, d, u32, u32, C.POINTER(Camera)]
        L.orc_gen_scene.restype = sz
        L.orc_gen_scene.argtypes = [u64, vp, sz]
        L.orc_trace.argtypes = [vp, i, vp, vp, sz, vp, vp, vp, vp]
        L.orc_generate_primary.argtypes = [C.POINTER(Camera), u64, i, u32, u32, u32, u32, vp, vp]
        L.orc_render.restype = d
        L.orc_render.argtypes = [vp, C.POINTER(Camera), i, i, u32, u32, u32, u64, i, u32, u32, u32, vp, vp, vp]
        L.orc_trace_paths.argtypes = [vp, C.POINTER(Camera), i, i, vp, vp, sz, u32, u64, vp, vp]
        L.orc_path_segments.argtypes = [vp, C.POINTER(Camera), i, i, vp, vp, sz, u32, u64, u32, vp, vp, vp]
        L.orc_finalize.argtypes = [vp, sz, d, vp]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_sphere_intersect.argtypes = [vp, d, vp, vp, i, vp]
        L.orc_disc_intersect.argtypes = [vp, vp, d, i, vp, vp, i, vp]
        L.orc_bbox_check.argtypes = [vp, vp, vp]
        L.orc_bbox_longest_axis.argtypes = [vp]
        L.orc_interval_intersection.argtypes = [d, d, d, d, vp]
        L.orc_interval_cover.argtypes = [d, d, d, d, vp]
        L.orc_tree_over_boxes.restype = sz
        L.orc_tree_over_boxes.argtypes = [vp, sz, vp, vp, sz]
        L.orc_sort_on_index.argtypes = [vp, sz, i, vp]
        L.orc_channel_to_ppm.restype = C.c_uint8
        L.orc_channel_to_ppm.argtypes = [d]
        L.orc_schlick.argtypes = [d, d, d]
        L.orc_scatter.argtypes = [vp, vp, vp, vp, i, u64, u32, u32, u32, vp, vp]
        L.orc_albedo.argtypes = [vp, i, vp, vp]
        L.orc_max_threads.restype = i
        L.orc_sizeof_hittable.restype = sz
        L.orc_sizeof_camera.restype = sz
        _lib = L
    return _lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def camera_build(lookat, lookfrom, inv_focal_length, aperture, horiz_res, vert_res):
    c = Camera()
    la, lf = (C.c_double * 3)(*map(float, lookat)), (C.c_double * 3)(*map(float, lookfrom))
    lib().orc_camera_build(la, lf, float(inv_focal_length), float(aperture), int(horiz_res), int(vert_res), C.byref(c))
    return c


def camera_from_bytes(raw):
    """Oracle camera from the bytes of the product's rt_camera (same layout)."""
    c = Camera()
    assert len(raw) == C.sizeof(Camera)
    C.memmove(C.byref(c), raw, len(raw))
    return c


def gen_scene(seed=1):
    n = lib().orc_gen_scene(int(seed), None, 0)
    out = np.zeros(n, dtype=HITTABLE_DTYPE)
    lib().orc_gen_scene(int(seed), out.ctypes.data, n)
    return out


class Scene:
    def __init__(self, hittables, textures=()):
        self.h = np.ascontiguousarray(hittables, dtype=HITTABLE_DTYPE)
        self._tex = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        tex = (Texture * max(1, len(self._tex)))()
        for k, t in enumerate(self._tex):
            tex[k].rgb8, tex[k].width, tex[k].height = t.ctypes.data, t.shape[1], t.shape[0]
        self.handle = C.c_void_p(lib().orc_scene_create(self.h.ctypes.data if len(self.h) else None, len(self.h), tex, len(self._tex)))
        self.tree_nodes = None

    def build_tree(self):
        self.tree_nodes = lib().orc_scene_build_tree(self.handle)
        return self.tree_nodes

    def tree_depth(self):
        return lib().orc_scene_tree_depth(self.handle)

    def set_tie_lowest_index(self, on):
        lib().orc_scene_set_tie_rule(self.handle, 1 if on else 0)

    def set_dir_noise(self, amplitude):
        """Test aid: move every ray direction by up to `amplitude` per component before tracing it (0 = off)."""
        lib().orc_scene_set_dir_noise(self.handle, float(amplitude))

    def trace(self, orig, dirs, traverse=TRAVERSE_GATED_BRUTE):
        o, d = f64(orig).reshape(-1, 3), f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        ids = np.full(n, -1, dtype=np.int32)
        t = np.zeros(n)
        nrm = np.zeros((n, 3))
        ctr = np.zeros(2, dtype=np.uint64)
        if traverse == TRAVERSE_TREE and self.tree_nodes is None:
            self.build_tree()
        lib().orc_trace(self.handle, traverse, o.ctypes.data, d.ctypes.data, n, ids.ctypes.data, t.ctypes.data, nrm.ctypes.data, ctr.ctypes.data)
        return ids, t, nrm, ctr

    def render(self, cam, spp, depth=10, seed=0, traverse=TRAVERSE_GATED_BRUTE, sampler=SAMPLER_DIRECT, sample_begin=0,
               threads=0, rows=None, row_stride=1):
        H, W = cam.vert_res, cam.horiz_res
        s = np.zeros((H, W, 3))
        q = np.zeros((H, W, 3), dtype=np.uint8)
        ctr = np.zeros(4, dtype=np.uint64)
        if traverse == TRAVERSE_TREE and self.tree_nodes is None:
            self.build_tree()
        r0, r1 = rows if rows else (0, H)
        th = threads if threads else lib().orc_max_threads()
        secs = lib().orc_render(self.handle, C.byref(cam), traverse, sampler, sample_begin, sample_begin + spp, depth, seed, th,
                                r0, r1, row_stride, s.ctypes.data, q.ctypes.data, ctr.ctypes.data)
        return s, q, {"seconds": secs, "box_tests": int(ctr[0]), "shape_tests": int(ctr[1]), "segments": int(ctr[2]), "paths": int(ctr[3]), "threads": th}

    def trace_paths(self, cam, pixel, sample, depth=10, seed=0, traverse=TRAVERSE_GATED_BRUTE, sampler=SAMPLER_DIRECT,
                    with_segments=False):
        p = np.ascontiguousarray(pixel, dtype=np.uint32)
        s = np.ascontiguousarray(sample, dtype=np.uint32)
        rgb = np.zeros((p.shape[0], 3))
        seg = np.zeros(p.shape[0], dtype=np.uint32)
        if traverse == TRAVERSE_TREE and self.tree_nodes is None:
            self.build_tree()
        lib().orc_trace_paths(self.handle, C.byref(cam), traverse, sampler, p.ctypes.data, s.ctypes.data, p.shape[0], depth, seed,
                              rgb.ctypes.data, seg.ctypes.data)
        return (rgb, seg) if with_segments else rgb

    def path_segments(self, cam, pixel, sample, depth=10, seed=0, traverse=TRAVERSE_GATED_BRUTE, sampler=SAMPLER_DIRECT,
                      max_segments=16):
        p = np.ascontiguousarray(pixel, dtype=np.uint32)
        s = np.ascontiguousarray(sample, dtype=np.uint32)
        n = p.shape[0]
        ids = np.full((n, max_segments), -1, dtype=np.int32)
        t = np.zeros((n, max_segments))
        n_seg = np.zeros(n, dtype=np.uint32)
        if traverse == TRAVERSE_TREE and self.tree_nodes is None:
            self.build_tree()
        lib().orc_path_segments(self.handle, C.byref(cam), traverse, sampler, p.ctypes.data, s.ctypes.data, n, depth, seed,
                                max_segments, ids.ctypes.data, t.ctypes.data, n_seg.ctypes.data)
        return ids, t, n_seg

    def albedo(self, index, loc):
        l = f64(loc)
        out = np.zeros(3)
        lib().orc_albedo(self.handle, int(index), l.ctypes.data, out.ctypes.data)
        return out

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_scene_destroy(self.handle)
            self.handle = None


def generate_primary(cam, seed, pixel_begin, pixel_end, sample_begin=0, sample_end=1, sampler=SAMPLER_DIRECT):
    n = (pixel_end - pixel_begin) * (sample_end - sample_begin)
    o, d = np.zeros((n, 3)), np.zeros((n, 3))
    lib().orc_generate_primary(C.byref(cam), seed, sampler, pixel_begin, pixel_end, sample_begin, sample_end, o.ctypes.data, d.ctypes.data)
    return o, d


def finalize(sum_rgb, spp):
    s = f64(sum_rgb)
    out = np.zeros(s.shape, dtype=np.uint8)
    lib().orc_finalize(s.ctypes.data, s.size, float(spp), out.ctypes.data)
    return out


def psnr(a, b, peak=255.0):
    mse = np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
