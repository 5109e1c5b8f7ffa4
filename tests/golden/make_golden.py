"""Regenerates the oracle-made golden vectors in this directory.

    python tests/golden/make_golden.py

The reference cannot be built here (Rust, no rustc), so these vectors come from
the CPU oracle, NOT from the reference; what pins the oracle to the reference is
reference_kats.json (the reference's own inline tests, restated as data).
scene.json and earthmap.jpeg are the reference's data files, copied verbatim;
earthmap_rgb8.npz is earthmap.jpeg decoded once with PIL (libjpeg-turbo) so that
oracle and device read the same texels wherever the tests run.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402


def edge_rays():
    """Hand-picked rays: axis-aligned (the 1e-4 divisor), origins inside spheres, grazing, pointing away."""
    o, d = [], []

    def add(orig, direction, normalise=True):
        v = np.asarray(direction, dtype=np.float64)
        if normalise:
            v = v / np.sqrt((v * v).sum())
        o.append(orig)
        d.append(v)

    add([13, 1.5, 3], [-1, 0, 0])           # one zero component pair: 1e-4 divisor on y and z
    add([0, 5, 0], [0, -1, 0])              # straight down on the glass sphere
    add([0, 1, 0], [0, 1, 0])               # from the centre of the glass sphere
    add([0, 1, 0], [1, 0.2, 0.1])
    add([4, 1, 0], [0, 0, 1])               # from inside the metal sphere
    add([-4, 1, 5], [0, 0, -1])
    add([0, 0.5, 20], [0, 0, -1])
    add([20, 0.2, 0.3], [-1, 0, 0])         # skims the row of small spheres
    add([0, 10, 0], [0, 1, 0])              # away from everything: miss
    add([0, -1000, 0], [0.3, 0.9, 0.1])     # from the centre of the ground sphere
    add([100, 50, 100], [-1, -0.5, -1])
    add([0, 2.0000001, -10], [0, 0, 1])     # grazes the top of the glass sphere
    add([0, 1.9999999, -10], [0, 0, 1])
    add([1.0, 1, -10], [0, 0, 1])           # tangent to the unit sphere in x
    add([0.99999, 1, -10], [0, 0, 1])
    add([3, 0.2, -20], [0, 0, 1])
    add([-11.5, 0.2, -11.5], [1, 0, 1])     # diagonal through the field at sphere height
    add([5, 3, 5], [-5, -2.8, -5])
    return np.array(o, dtype=np.float64), np.array(d, dtype=np.float64)


def main():
    from PIL import Image

    with Image.open(os.path.join(HERE, "earthmap.jpeg")) as im:
        earth = np.asarray(im.convert("RGB"), dtype=np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "earthmap_rgb8.npz"), rgb8=earth)

    scene_h = ob.gen_scene(1)
    sha = hashlib.sha256(scene_h.tobytes()).hexdigest()
    scene = ob.Scene(scene_h)
    assert scene.build_tree() == 975

    # ---- fixed primary-ray dump: camera rays (f32-representable) + edge cases
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 96, 64)
    o, d = ob.generate_primary(cam, seed=0, pixel_begin=0, pixel_end=96 * 64)
    eo, ed = edge_rays()
    o32 = np.concatenate([o, eo]).astype(np.float32)
    d32 = np.concatenate([d, ed]).astype(np.float32)
    ids, t, nrm, _ = scene.trace(o32, d32, ob.TRAVERSE_GATED_BRUTE)
    ids_t, t_t, nrm_t, _ = scene.trace(o32, d32, ob.TRAVERSE_TREE)
    assert np.array_equal(ids, ids_t) and np.array_equal(t[ids >= 0], t_t[ids >= 0]), "tree and topology-free statement disagree"
    np.savez_compressed(os.path.join(HERE, "primary_rays_seed1.npz"), orig=o32, dir=d32, id=ids, t=t, normal=nrm,
                        scene_sha256=np.array(sha))
    print("primary rays: %d, hits %d, distinct ids %d" % (len(ids), (ids >= 0).sum(), len(np.unique(ids))))

    # ---- small renders of the seed-1 scene (sum images, f64)
    cam_r = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 48, 32)
    s_direct, _, st = scene.render(cam_r, spp=8, depth=10, seed=0, sampler=ob.SAMPLER_DIRECT)
    s_ref, _, _ = scene.render(cam_r, spp=8, depth=10, seed=0, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    np.savez_compressed(os.path.join(HERE, "render_seed1_48x32_8spp.npz"), sum_direct=s_direct, sum_reference=s_ref,
                        segments_per_path=np.array(st["segments"] / st["paths"]))
    print("render: segments/path %.3f" % (st["segments"] / st["paths"]))

    # ---- bundled scene.json through the linear path, texture lookups included
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    rng = np.random.default_rng(5)
    v = rng.normal(size=(256, 3))
    v /= np.sqrt((v * v).sum(axis=1, keepdims=True))
    tex_h = np.zeros(1, dtype=ob.HITTABLE_DTYPE)
    tex_h["shape"], tex_h["material"], tex_h["texture"] = 0, 3, 0
    tex_h["centre"], tex_h["radius"] = [0.0, 0.2, 3.0], 0.75
    tex_h["orient_up"], tex_h["orient_around"] = [0.0, 1.0, -0.6], [1.0, 0.0, 0.5]
    tex_scene = ob.Scene(tex_h, [earth])
    alb = np.array([tex_scene.albedo(0, x) for x in v])
    np.savez_compressed(os.path.join(HERE, "earth_albedo_lookups.npz"), loc=v, albedo=alb)
    print("texture lookups: mean albedo", alb.mean(axis=0))


if __name__ == "__main__":
    main()
