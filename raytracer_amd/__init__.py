"""MI355X-native backend for the render path of nabilbrice/raytracer.

Everything computes behind the C ABI of ``include/rt_mi355x.h`` (hand-written
HIP for gfx950); this package is the thin host mirror of the crate's interface.
"""
from ._abi import (HITTABLE_DTYPE, LIB_PATH, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE, RT_MAT_EMITTER, RT_MAT_METAL,
                   RT_MAT_TEXTUREMAP, RT_SHAPE_DISC, RT_SHAPE_SPHERE, RtError)
from .api import (REFERENCE_DEPTH, Camera, Config, Scene, accel_render_into_file, decode_jpeg, default_camera,
                  dielectric, diffuse, disc, emitter, gen_scene, gen_scene_grid, gen_scene_shells, hittables, load_jpeg, metal, ppm_header,
                  render_into_file, render_multi, sphere, texture_map, write_ppm, write_ppm_p6)
from .shard import render_step, sample_shard

__all__ = [n for n in dir() if not n.startswith("_")]
