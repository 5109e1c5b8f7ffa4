"""Frozen work-model constants shared by bench.py and the written reports.

Algorithmic flops are counted on the REFERENCE's arithmetic (add/sub/mul/div/sqrt/compare-select = 1),
with a balanced-tree traversal model: a segment tests 2*ceil(log2 N) boxes and one sphere.
"""
import math

F_BOX = 24        # src/boundingvolume.rs:70-95: 9 sub, 6 div, 9 compare/select
F_SPHERE = 23     # src/geometry.rs:104-121
F_CAMERA = 88     # src/camera.rs:49-78 + src/ray.rs:9-11
F_SHADE = 90      # per segment: hit point, normal, scatter (80 % diffuse 87 / 15 % metal 110 / 5 % glass 80), albedo, Ray::new
F_ACCUMULATE = 3  # per path
NODE_BYTES = 32   # one box = 6 f32 + 2 indices
SPHERE_BYTES = 16

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 (packed v_pk_fma_f32 rate); the non-packed ceiling is 78.6
PEAK_HBM_GBS = 8000.0


def flops_per_segment(n_spheres):
    return 2 * math.ceil(math.log2(max(n_spheres, 2))) * F_BOX + F_SPHERE + F_SHADE


def flops_per_path(n_spheres, segments_per_path):
    return F_CAMERA + segments_per_path * flops_per_segment(n_spheres) + F_ACCUMULATE


def bytes_per_segment(n_spheres):
    return 2 * math.ceil(math.log2(max(n_spheres, 2))) * NODE_BYTES + SPHERE_BYTES


# Mean segments per path of the bench workload (random spheres seed 1, default camera, depth 50), as counted by the
# CPU oracle in the reference's configuration; bench.py uses it when its CPU leg is skipped (N > 1, --cpu-seconds 0).
C2_SEGMENTS_PER_PATH = 2.0263
