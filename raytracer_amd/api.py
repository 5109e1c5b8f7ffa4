"""Host-side mirror of the reference's library interface over the C ABI.

Names follow the crate: ``Camera.build`` (src/camera.rs:19), ``gen_scene`` /
``default_camera`` (src/scenegen.rs:12, :70), ``Config`` (src/config.rs:7-28),
``render_into_file`` / ``accel_render_into_file`` (src/lib.rs:137, :159).  All
arithmetic happens behind ``include/rt_mi355x.h``; this file only marshals.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import HITTABLE_DTYPE, RtError, check, lib, ptr

REFERENCE_DEPTH = 10  # the literal the reference passes, src/lib.rs:145 and :172


def _vec3(v):
    a = (C.c_double * 3)(*[float(x) for x in v])
    return a


class Camera:
    """src/camera.rs:4-16."""

    def __init__(self, c_struct):
        self.c = c_struct

    @staticmethod
    def build(lookat, lookfrom, inv_focal_length, aperture, horiz_res, vert_res):
        cam = _abi.rt_camera()
        check(lib.rt_camera_build(_vec3(lookat), _vec3(lookfrom), float(inv_focal_length), float(aperture),
                                  int(horiz_res), int(vert_res), C.byref(cam)))
        return Camera(cam)

    horiz_res = property(lambda self: self.c.horiz_res)
    vert_res = property(lambda self: self.c.vert_res)

    def field(self, name):
        v = getattr(self.c, name)
        return np.array(list(v)) if hasattr(v, "__len__") else v


def default_camera(horiz_res=512, vert_res=512):
    """src/scenegen.rs:70-79 (the reference fixes 512 x 512)."""
    cam = _abi.rt_camera()
    check(lib.rt_default_camera(int(horiz_res), int(vert_res), C.byref(cam)))
    return Camera(cam)


def _generated(fn, *args):
    n = C.c_size_t(0)
    check(fn(*args, None, 0, C.byref(n)))
    out = np.zeros(n.value, dtype=HITTABLE_DTYPE)
    check(fn(*args, ptr(out), n.value, C.byref(n)))
    return out


def gen_scene(seed=1):
    """src/scenegen.rs:12-68 as an array of hittables in push order (488 of them)."""
    return _generated(lib.rt_gen_scene, int(seed))


def gen_scene_grid(seed, side):
    return _generated(lib.rt_gen_scene_grid, int(seed), int(side))


def gen_scene_shells(shells):
    return _generated(lib.rt_gen_scene_shells, int(shells))


# --- hittable constructors (src/geometry.rs:36-47, src/materials.rs:15-35) ---
def _blank():
    h = np.zeros((), dtype=HITTABLE_DTYPE)
    h["texture"] = -1
    return h


def diffuse(albedo):
    return {"material": _abi.RT_MAT_DIFFUSE, "albedo": albedo}


def metal(albedo, fuzz):
    return {"material": _abi.RT_MAT_METAL, "albedo": albedo, "fuzz": fuzz}


def dielectric(refractive_index):
    return {"material": _abi.RT_MAT_DIELECTRIC, "refractive_index": refractive_index}


def texture_map(texture, orient_up, orient_around):
    return {"material": _abi.RT_MAT_TEXTUREMAP, "texture": texture, "orient_up": orient_up, "orient_around": orient_around}


def emitter(albedo):
    return {"material": _abi.RT_MAT_EMITTER, "albedo": albedo}


def sphere(centre, radius, material):
    h = _blank()
    h["shape"] = _abi.RT_SHAPE_SPHERE
    h["centre"] = centre
    h["radius"] = radius
    for k, v in material.items():
        h[k] = v
    return h


def disc(centre, normal, radius, material, normalise=True):
    """Disc::new normalises the normal (src/geometry.rs:129-131); serde does not."""
    h = _blank()
    h["shape"] = _abi.RT_SHAPE_DISC
    h["centre"] = centre
    n = np.asarray(normal, dtype=np.float64)
    h["normal"] = n / np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) if normalise else n
    h["radius"] = radius
    for k, v in material.items():
        h[k] = v
    return h


def hittables(items):
    return np.array(items, dtype=HITTABLE_DTYPE) if len(items) else np.zeros(0, dtype=HITTABLE_DTYPE)


class Scene:
    """Owner of an rt_scene handle (Box<[Hittable]> / Box<CoveringTree>, src/main.rs:10-13)."""

    def __init__(self, hittable_array, textures=(), device=0, builder="sah"):
        """builder: "sah" (host, binned surface-area heuristic) or "lbvh" (built on the GPU from Morton codes)."""
        h = np.ascontiguousarray(hittable_array, dtype=HITTABLE_DTYPE)
        self.hittables = h
        self._tex_keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        for t in self._tex_keep:
            if t.ndim != 3 or t.shape[2] != 3:
                raise ValueError("a texture is an (H, W, 3) uint8 array")
        tex = (_abi.rt_texture * max(1, len(self._tex_keep)))()
        for i, t in enumerate(self._tex_keep):
            tex[i].rgb8 = t.ctypes.data
            tex[i].width = t.shape[1]
            tex[i].height = t.shape[0]
        handle = C.c_void_p(None)
        so = _abi.rt_scene_opts()
        so.builder = {"sah": _abi.RT_BUILDER_SAH_HOST, "lbvh": _abi.RT_BUILDER_LBVH_DEVICE}[builder]
        check(lib.rt_scene_create_ex(ptr(h) if len(h) else None, len(h), tex, len(self._tex_keep), int(device), C.byref(so),
                                     C.byref(handle)))
        self.handle = handle
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            lib.rt_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    @property
    def info(self):
        i = _abi.rt_scene_info()
        check(lib.rt_scene_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    @staticmethod
    def opts(sample_begin, sample_end, depth, seed, flags=0, samples_per_item=0):
        o = _abi.rt_render_opts()
        o.sample_begin, o.sample_end, o.depth, o.flags, o.seed = int(sample_begin), int(sample_end), int(depth), int(flags), int(seed)
        o.samples_per_item = int(samples_per_item)
        return o

    def render_chunk(self, cam, spp, sample_begin=0, samples_per_item=0):
        """Samples per work item (the image's summation order) for this scene, frame and sample count."""
        o = Scene.opts(sample_begin, sample_begin + spp, 1, 0, 0, samples_per_item)
        out = C.c_uint32(0)
        check(lib.rt_render_chunk(self.handle, C.byref(cam.c), C.byref(o), C.byref(out)))
        return out.value

    def render(self, cam, spp, depth=REFERENCE_DEPTH, seed=0, sample_begin=0, want_sum=True, want_rgb8=True, flags=0,
               samples_per_item=0):
        """Host-buffer render: returns (sum_rgb float32 (H,W,3) or None, rgb8 (H,W,3) or None)."""
        H, W = cam.vert_res, cam.horiz_res
        s = np.zeros((H, W, 3), dtype=np.float32) if want_sum else None
        q = np.zeros((H, W, 3), dtype=np.uint8) if want_rgb8 else None
        o = self.opts(sample_begin, sample_begin + spp, depth, seed, flags, samples_per_item)
        check(lib.rt_render(self.handle, C.byref(cam.c), C.byref(o), ptr(s), ptr(q)))
        return s, q

    def render_device(self, cam, d_sum_ptr, sample_begin, sample_end, depth, seed, stream=None, accumulate=False, flags=0,
                      samples_per_item=0):
        """Asynchronous render into device memory (d_sum_ptr: integer address of W*H*3 floats)."""
        o = self.opts(sample_begin, sample_end, depth, seed, flags | (_abi.RT_RENDER_ACCUMULATE if accumulate else 0), samples_per_item)
        check(lib.rt_render_device(self.handle, C.byref(cam.c), C.byref(o), C.c_void_p(d_sum_ptr), C.c_void_p(stream or 0)))

    def reserve(self, cam, spp, sample_begin=0, samples_per_item=0):
        """Sizes the partial-sum buffer ahead of time, so that render_device only enqueues."""
        o = self.opts(sample_begin, sample_begin + spp, 1, 0, 0, samples_per_item)
        check(lib.rt_scene_reserve(self.handle, C.byref(cam.c), C.byref(o)))

    def render_paths(self, cam, spp, depth=REFERENCE_DEPTH, seed=0, sample_begin=0, flags=0, samples_per_item=0):
        """Colour of every path of a render, from the render kernel itself: float32 (H*W, spp, 3)."""
        n = cam.vert_res * cam.horiz_res
        rgb = np.zeros((n, spp, 3), dtype=np.float32)
        o = self.opts(sample_begin, sample_begin + spp, depth, seed, flags, samples_per_item)
        check(lib.rt_debug_render_paths(self.handle, C.byref(cam.c), C.byref(o), ptr(rgb)))
        return rgb

    def finalize_device(self, d_sum_ptr, n_pixels, spp_total, d_rgb8_ptr, stream=None):
        check(lib.rt_finalize_device(self.handle, C.c_void_p(d_sum_ptr), int(n_pixels), int(spp_total),
                                     C.c_void_p(d_rgb8_ptr), C.c_void_p(stream or 0)))

    def trace_primary(self, orig, dirs):
        """Nearest hit per ray: (id int32, t float64, normal float64 (n,3)); id = -1 on a miss."""
        o = np.ascontiguousarray(orig, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("orig and dir must have the same shape")
        n = o.shape[0]
        ids = np.full(n, -1, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        nrm = np.zeros((n, 3), dtype=np.float64)
        check(lib.rt_trace_primary(self.handle, ptr(o), ptr(d), n, ptr(ids), ptr(t), ptr(nrm)))
        return ids, t, nrm

    def generate_primary(self, cam, seed, pixel_begin, pixel_end, sample_begin=0, sample_end=1):
        n = (pixel_end - pixel_begin) * (sample_end - sample_begin)
        o = np.zeros((n, 3), dtype=np.float32)
        d = np.zeros((n, 3), dtype=np.float32)
        check(lib.rt_generate_primary(self.handle, C.byref(cam.c), int(seed), int(pixel_begin), int(pixel_end),
                                      int(sample_begin), int(sample_end), ptr(o), ptr(d)))
        return o, d

    def trace_paths(self, cam, pixel, sample, depth=REFERENCE_DEPTH, seed=0):
        p = np.ascontiguousarray(pixel, dtype=np.uint32)
        s = np.ascontiguousarray(sample, dtype=np.uint32)
        rgb = np.zeros((p.shape[0], 3), dtype=np.float32)
        o = self.opts(0, 1, depth, seed)
        check(lib.rt_trace_paths(self.handle, C.byref(cam.c), C.byref(o), ptr(p), ptr(s), p.shape[0], ptr(rgb)))
        return rgb


def _path_segments(self, cam, pixel, sample, depth=REFERENCE_DEPTH, seed=0, max_segments=16):
    """Per path: (ids (n,max_segments) int32, -1 = the closing miss / unused; t float64; n_seg uint32)."""
    p = np.ascontiguousarray(pixel, dtype=np.uint32)
    s = np.ascontiguousarray(sample, dtype=np.uint32)
    n = p.shape[0]
    ids = np.full((n, max_segments), -1, dtype=np.int32)
    t = np.zeros((n, max_segments), dtype=np.float64)
    n_seg = np.zeros(n, dtype=np.uint32)
    o = self.opts(0, 1, depth, seed)
    check(lib.rt_trace_path_segments(self.handle, C.byref(cam.c), C.byref(o), ptr(p), ptr(s), n, None, int(max_segments),
                                     ptr(ids), ptr(t), ptr(n_seg)))
    return ids, t, n_seg


Scene.path_segments = _path_segments


def render_multi(scenes, cam, spp, depth=REFERENCE_DEPTH, seed=0, sample_begin=0, want_sum=True, want_rgb8=True, samples_per_item=0):
    """One render split by sample index over the devices of `scenes` (one Scene per device, same hittables), summed
    on the first one: returns (sum_rgb, rgb8) like Scene.render."""
    H, W = cam.vert_res, cam.horiz_res
    s = np.zeros((H, W, 3), dtype=np.float32) if want_sum else None
    q = np.zeros((H, W, 3), dtype=np.uint8) if want_rgb8 else None
    o = Scene.opts(sample_begin, sample_begin + spp, depth, seed, 0, samples_per_item)
    handles = (C.c_void_p * len(scenes))(*[sc.handle for sc in scenes])
    check(lib.rt_render_multi(handles, len(scenes), C.byref(cam.c), C.byref(o), ptr(s), ptr(q)))
    return s, q


def write_ppm(path, rgb8):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    check(lib.rt_write_ppm(os.fsencode(path), ptr(a), a.shape[1], a.shape[0]))


def write_ppm_p6(path, rgb8):
    """Binary PPM: the header of src/main.rs:41 with the magic number P6, then the raw bytes."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    check(lib.rt_write_ppm_p6(os.fsencode(path), ptr(a), a.shape[1], a.shape[0]))


def ppm_header(cam):
    """src/main.rs:41."""
    return "P3\n%d %d\n255\n" % (cam.horiz_res, cam.vert_res)


def _write_pixel_lines(file, rgb8):
    # src/lib.rs:151: one "r g b\n" line per pixel, rows top to bottom
    flat = rgb8.reshape(-1, 3)
    file.write("\n".join("%d %d %d" % (p[0], p[1], p[2]) for p in flat.tolist()))
    file.write("\n")


def accel_render_into_file(file, cam, scene, spp, depth=REFERENCE_DEPTH, seed=0):
    """src/lib.rs:159-184: `file` is an open text file positioned after the header.

    Like the reference's tree builder (src/boundingvolume.rs:254-256) this entry
    point accepts spheres only.
    """
    if scene.info["n_discs"]:
        raise RtError(_abi.RT_ERR_UNSUPPORTED, "accel_render_into_file: the covering tree holds spheres only")
    _, rgb8 = scene.render(cam, spp, depth=depth, seed=seed, want_sum=False)
    _write_pixel_lines(file, rgb8)


def render_into_file(file, cam, scene, spp, depth=REFERENCE_DEPTH, seed=0):
    """src/lib.rs:137-157: any mix of spheres and discs."""
    _, rgb8 = scene.render(cam, spp, depth=depth, seed=seed, want_sum=False)
    _write_pixel_lines(file, rgb8)


class Config:
    """src/config.rs:7-28 + the serde schema of Shape / Material / Color / Vec3, read by the library
    (rt_config_load / rt_config_parse).

    ``"map"`` of a TextureMap is a PATH, decoded while the scene is read (src/materials.rs:37-46);
    a relative path is resolved against the scene file's directory.
    """

    def __init__(self, camera, hittable_array, textures):
        self.camera = camera
        self.hittables = hittable_array
        self.textures = textures

    @classmethod
    def _from_handle(cls, handle):
        try:
            cam = _abi.rt_camera()
            hp, tp = C.c_void_p(), C.c_void_p()
            n, nt = C.c_size_t(0), C.c_size_t(0)
            check(lib.rt_config_view(handle, C.byref(cam), C.byref(hp), C.byref(n), C.byref(tp), C.byref(nt)))
            items = np.zeros(n.value, dtype=_abi.HITTABLE_DTYPE)
            if n.value:
                C.memmove(items.ctypes.data, hp.value, items.nbytes)
            textures = []
            tex = C.cast(tp, C.POINTER(_abi.rt_texture))
            for i in range(nt.value):
                t = tex[i]
                buf = (C.c_uint8 * (t.width * t.height * 3)).from_address(t.rgb8)
                textures.append(np.frombuffer(buf, dtype=np.uint8).reshape(t.height, t.width, 3).copy())
            return cls(Camera(cam), items, textures)
        finally:
            lib.rt_config_destroy(handle)

    @classmethod
    def from_json(cls, path):
        handle = C.c_void_p()
        check(lib.rt_config_load(os.fsencode(path), C.byref(handle)))
        return cls._from_handle(handle)

    @classmethod
    def from_text(cls, text, texture_dir=""):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        handle = C.c_void_p()
        check(lib.rt_config_parse(raw, len(raw), os.fsencode(texture_dir), C.byref(handle)))
        return cls._from_handle(handle)


def load_jpeg(path):
    """image::open of src/materials.rs:37-39 for baseline JPEG files: (H, W, 3) uint8."""
    p, w, h = C.c_void_p(), C.c_uint32(0), C.c_uint32(0)
    check(lib.rt_load_jpeg(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    try:
        buf = (C.c_uint8 * (w.value * h.value * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, w.value, 3).copy()
    finally:
        lib.rt_free(p)


def decode_jpeg(data):
    raw = bytes(data)
    p, w, h = C.c_void_p(), C.c_uint32(0), C.c_uint32(0)
    check(lib.rt_decode_jpeg(raw, len(raw), C.byref(p), C.byref(w), C.byref(h)))
    try:
        buf = (C.c_uint8 * (w.value * h.value * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, w.value, 3).copy()
    finally:
        lib.rt_free(p)
