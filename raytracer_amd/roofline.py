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

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32: 1024 SIMDs x 32 lanes x 2 flop (plain v_fma_f32, one wave64 instruction per 2 cycles) x 2.4 GHz
PEAK_HBM_GBS = 8000.0

# What binds the descent of a tree that lives in HBM / the caches (1M spheres and up) is not HBM bandwidth — the nodes
# a frame touches sit in L1 / L2 / Infinity Cache — but the rate at which a CU's vector memory path serves per-lane
# gathers: every lane of a wave fetches its own 32-byte node (two 16-byte loads) per step.  Measured on one MI355X with
# tools/microbench/gather_rate.hip (dependent random walks, 16 waves per CU, all 64 lanes on), in 1e9 node fetches
# per second over the chip, by where the table lives:
GATHER_GNODES_L1 = 925.0      # 32 KB table: every fetch hits the CU's L1 — the ceiling of the gather path itself
GATHER_GNODES_L2 = 410.0      # 4 MB table: the XCD's L2
GATHER_GNODES_MALL = 250.0    # 32 MB table: L2 misses served by the Infinity Cache (the rate does not depend on lanes on or waves)
GATHER_GNODES_HBM = 195.0     # 256 MB table: beyond the Infinity Cache


def flops_per_segment(n_spheres):
    return 2 * math.ceil(math.log2(max(n_spheres, 2))) * F_BOX + F_SPHERE + F_SHADE


def flops_per_path(n_spheres, segments_per_path):
    return F_CAMERA + segments_per_path * flops_per_segment(n_spheres) + F_ACCUMULATE


def bytes_per_segment(n_spheres):
    return 2 * math.ceil(math.log2(max(n_spheres, 2))) * NODE_BYTES + SPHERE_BYTES


def measured_traffic_bytes(summary_path=None):
    """HBM bytes per launch of the bench workload's render kernel (+ combine_kernel), from the committed summary of
    separate `rocprofv3 --pmc` passes over `bench.py --steps 1 --warmup 0` (scripts/pmc_passes.sh): FETCH_SIZE and
    WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes read, hence the factor 2.  Returns
    {"bytes_per_launch", "source"} or None when no summary is there."""
    import glob
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = [summary_path] if summary_path else sorted(glob.glob(os.path.join(root, "profiles", "r*", "bench_pmc_summary.txt")))
    for path in reversed(paths):
        vals = {}
        try:
            for line in open(path):
                parts = line.split()
                if len(parts) == 2:
                    vals[parts[0]] = float(parts[1])
        except OSError:
            continue
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return {"bytes_per_launch": int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
                    "source": os.path.relpath(path, root) + ": (2 x FETCH_SIZE + WRITE_SIZE) KiB, one launch"}
    return None


# Mean segments per path of the bench workload (random spheres seed 1, default camera, depth 50), as counted by the
# CPU oracle in the reference's configuration; bench.py uses it when its CPU leg is skipped (N > 1, --cpu-seconds 0).
C2_SEGMENTS_PER_PATH = 2.0263
