"""ctypes binding of include/rt_mi355x.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
gfx950).  There is no CPU fallback: importing this module without the built
library raises, and every device entry point fails with RT_ERR_DEVICE when no
gfx950 GPU is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "lib", "librt_mi355x.so")  # the variable: another build of the library, for A/B measurements

RT_OK = 0
RT_ERR_INVALID_ARGUMENT, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_IO, RT_ERR_LIMIT = 1, 2, 3, 4, 5
RT_SHAPE_SPHERE, RT_SHAPE_DISC = 0, 1
RT_MAT_DIFFUSE, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_TEXTUREMAP, RT_MAT_EMITTER = 0, 1, 2, 3, 4
RT_RENDER_ACCUMULATE, RT_RENDER_KERNEL_LOCKSTEP, RT_RENDER_KERNEL_NO_LDS_SCENE = 1, 2, 4
RT_RENDER_STATS, RT_RENDER_TIMING, RT_RENDER_DEFER_LEAF, RT_RENDER_KERNEL_WAVEFRONT = 8, 16, 32, 64
RT_NOALLOC_SCRATCH_BYTES = 64 * 1024 * 1024

# rt_hittable as a numpy record (same layout as the C struct), so that a million
# hittables are one array and not a million Python objects.
HITTABLE_DTYPE = np.dtype(
    [
        ("shape", "<i4"),
        ("material", "<i4"),
        ("centre", "<f8", (3,)),
        ("normal", "<f8", (3,)),
        ("radius", "<f8"),
        ("albedo", "<f8", (3,)),
        ("fuzz", "<f8"),
        ("refractive_index", "<f8"),
        ("orient_up", "<f8", (3,)),
        ("orient_around", "<f8", (3,)),
        ("texture", "<i4"),
        ("reserved", "<i4"),
    ],
    align=True,
)


class rt_camera(C.Structure):
    _fields_ = [
        ("lookat", C.c_double * 3),
        ("lookfrom", C.c_double * 3),
        ("lookup", C.c_double * 3),
        ("focal_distance", C.c_double),
        ("inv_focal_length", C.c_double),
        ("aperture", C.c_double),
        ("horiz_arm", C.c_double * 3),
        ("vert_arm", C.c_double * 3),
        ("horiz_res", C.c_uint32),
        ("vert_res", C.c_uint32),
        ("aspect_ratio", C.c_double),
    ]


class rt_texture(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class rt_scene_info(C.Structure):
    _fields_ = [
        ("n_hittables", C.c_uint64),
        ("n_spheres", C.c_uint64),
        ("n_discs", C.c_uint64),
        ("n_nodes", C.c_uint64),
        ("bvh_depth", C.c_uint32),
        ("stack_depth", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("build_seconds", C.c_double),
        ("device", C.c_int32),
        ("lds_resident", C.c_int32),
        ("n_wide", C.c_uint32),
        ("n_shells", C.c_uint32),
    ]


class rt_checkpoint_info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_done", C.c_uint32), ("depth", C.c_uint32),
                ("seed", C.c_uint64), ("tag", C.c_uint64)]


class rt_scene_opts(C.Structure):
    _fields_ = [("builder", C.c_uint32), ("reserved", C.c_uint32 * 7)]


RT_BUILDER_DEFAULT, RT_BUILDER_SAH_HOST, RT_BUILDER_LBVH_DEVICE = 0, 1, 2


class rt_render_opts(C.Structure):
    _fields_ = [
        ("sample_begin", C.c_uint32),
        ("sample_end", C.c_uint32),
        ("depth", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("samples_per_item", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


_vp, _sz, _u32, _u64, _i, _d = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_double
_dp3 = C.POINTER(C.c_double)

# name -> (restype, argtypes).  Every function declared in include/rt_mi355x.h.
PROTOTYPES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (_u32, []),
    "rt_device_count": (_i, []),
    "rt_camera_build": (_i, [_dp3, _dp3, _d, _d, _u32, _u32, C.POINTER(rt_camera)]),
    "rt_default_camera": (_i, [_u32, _u32, C.POINTER(rt_camera)]),
    "rt_gen_scene": (_i, [_u64, _vp, _sz, C.POINTER(_sz)]),
    "rt_gen_scene_grid": (_i, [_u64, _u32, _vp, _sz, C.POINTER(_sz)]),
    "rt_gen_scene_shells": (_i, [_u32, _vp, _sz, C.POINTER(_sz)]),
    "rt_scene_create": (_i, [_vp, _sz, C.POINTER(rt_texture), _sz, _i, C.POINTER(_vp)]),
    "rt_scene_create_ex": (_i, [_vp, _sz, C.POINTER(rt_texture), _sz, _i, _vp, C.POINTER(_vp)]),
    "rt_scene_destroy": (None, [_vp]),
    "rt_scene_get_info": (_i, [_vp, C.POINTER(rt_scene_info)]),
    "rt_render_device": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp]),
    "rt_render_device_noalloc": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp]),
    "rt_scene_reserve": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts)]),
    "rt_debug_render_stats": (_i, [_vp, _vp]),
    "rt_debug_render_paths": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp]),
    "rt_debug_render_launch": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_render_chunk": (_i, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_render_opts), C.POINTER(_u32)]),
    "rt_finalize_device": (_i, [_vp, _vp, _sz, _u32, _vp, _vp]),
    "rt_render": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp]),
    "rt_render_multi": (_i, [C.POINTER(_vp), _sz, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp]),
    "rt_trace_primary": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "rt_generate_primary": (_i, [_vp, C.POINTER(rt_camera), _u64, _u32, _u32, _u32, _u32, _vp, _vp]),
    "rt_trace_paths": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp, _sz, _vp]),
    "rt_trace_path_segments": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), _vp, _vp, _sz, _vp, _u32, _vp, _vp, _vp]),
    "rt_write_ppm": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "rt_write_ppm_p6": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "rt_finalize": (_i, [_vp, _sz, _u32, _vp]),
    "rt_checkpoint_save": (_i, [C.c_char_p, _vp, _vp]),
    "rt_checkpoint_load": (_i, [C.c_char_p, _vp, _vp, _sz]),
    "rt_hash_bytes": (_u64, [_vp, _sz, _u64]),
    "rt_config_load": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "rt_config_parse": (_i, [C.c_char_p, _sz, C.c_char_p, C.POINTER(_vp)]),
    "rt_config_destroy": (None, [_vp]),
    "rt_config_view": (_i, [_vp, C.POINTER(rt_camera), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)]),
    "rt_decode_jpeg": (_i, [_vp, _sz, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_load_jpeg": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_free": (None, [_vp]),
    "rt_render_into_file": (_i, [_i, C.POINTER(rt_camera), _vp, _sz, C.POINTER(rt_texture), _sz, _u32, _u32, _u64, _i]),
    "rt_accel_render_into_file": (_i, [_i, C.POINTER(rt_camera), _vp, _sz, C.POINTER(rt_texture), _sz, _u32, _u32, _u64, _i]),
    "rt_render_into_file_p6": (_i, [_i, C.POINTER(rt_camera), _vp, _sz, C.POINTER(rt_texture), _sz, _u32, _u32, _u64, _i]),
    "rt_accel_render_into_file_p6": (_i, [_i, C.POINTER(rt_camera), _vp, _sz, C.POINTER(rt_texture), _sz, _u32, _u32, _u64, _i]),
}


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rt_mi355x status %d: %s" % (status, message))
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status):
    if status != RT_OK:
        raise RtError(status, lib.rt_last_error().decode("utf-8", "replace"))


def ptr(a):
    """Address of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
