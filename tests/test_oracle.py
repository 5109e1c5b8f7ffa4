"""CPU: the oracle against the reference's own known-answer tests and its committed golden vectors."""
import hashlib
import json
import os
import subprocess

import numpy as np

from conftest import GOLDEN, golden


def test_selftest_binary_passes(ob):
    # the reference's 29 inline tests restated in C++, plus hand-derived checks for the unpinned rules
    out = subprocess.run([ob.ORACLE_SELFTEST], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_struct_layouts(ob):
    import ctypes

    assert ob.lib().orc_sizeof_hittable() == ob.HITTABLE_DTYPE.itemsize == 160
    assert ob.lib().orc_sizeof_camera() == ctypes.sizeof(ob.Camera)


def _kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


_KEEP = []


def _arr(x):
    # the returned array must outlive the ctypes call that receives its address
    a = np.ascontiguousarray(x, dtype=np.float64)
    _KEEP.append(a)
    return a


def test_reference_kats_shapes(ob):
    L = ob.lib()
    k = _kats()
    t = np.zeros(1)
    for c in k["sphere_intersect"]:
        hit = L.orc_sphere_intersect(_arr(c["centre"]).ctypes.data, c["radius"], _arr(c["orig"]).ctypes.data,
                                     _arr(c["dir"]).ctypes.data, int(c["normalise_dir"]), t.ctypes.data)
        if c["t"] == "none":
            assert hit == 0, c["cite"]
        else:
            assert hit == 1 and t[0] == c["t"], c["cite"]  # exact, as assert_eq! in the reference
    for c in k["disc_intersect"]:
        hit = L.orc_disc_intersect(_arr(c["centre"]).ctypes.data, _arr(c["normal"]).ctypes.data, c["radius"], 1,
                                   _arr(c["orig"]).ctypes.data, _arr(c["dir"]).ctypes.data, 1, t.ctypes.data)
        assert hit == 1
        d = _arr(c["dir"]) / np.linalg.norm(c["dir"])
        assert np.array_equal(_arr(c["orig"]) + t[0] * d, _arr(c["position"])), c["cite"]


def test_reference_kats_boxes_and_intervals(ob):
    L = ob.lib()
    k = _kats()
    for c in k["bbox_check_intersection"]:
        got = L.orc_bbox_check(_arr(c["dims"]).ctypes.data, _arr(c["orig"]).ctypes.data, _arr(c["dir"]).ctypes.data)
        assert bool(got) == c["hit"], c["cite"]
    out = np.zeros(2)
    for c in k["interval_intersection"]:
        got = L.orc_interval_intersection(c["a"][0], c["a"][1], c["b"][0], c["b"][1], out.ctypes.data)
        assert (got == 0) == (c["result"] == "none"), c["cite"]
    for c in k["interval_cover"]:
        L.orc_interval_cover(c["a"][0], c["a"][1], c["b"][0], c["b"][1], out.ctypes.data)
        assert list(out) == c["result"], c["cite"]
    for c in k["longest_axis"]:
        assert L.orc_bbox_longest_axis(_arr(c["dims"]).ctypes.data) == c["axis"], c["cite"]
    for c in k["sort_on_index"]:
        b = _arr(c["boxes"])
        order = np.zeros(len(b), dtype=np.int32)
        L.orc_sort_on_index(b.ctypes.data, len(b), c["axis"], order.ctypes.data)
        assert order[0] == c["first"], c["cite"]
    for c in k["covering_tree"]:
        b = _arr(c["boxes"])
        covers = np.zeros((16, 6))
        boxed = np.zeros(16, dtype=np.int32)
        n = L.orc_tree_over_boxes(b.ctypes.data, len(b), covers.ctypes.data, boxed.ctypes.data, 16)
        assert n == c["n_nodes"]
        # pre-order: root, left subtree (one leaf), right inner node, its two leaves
        assert list(covers[0]) == c["root_cover"]
        assert boxed[1] == c["left_boxed"]
        assert boxed[2] == -1 and list(covers[2]) == c["right_cover"], c["cite"]


def test_reference_kats_tree_traversal(ob):
    k = _kats()
    for name in ("coveringtree_intersect", "debug_scene"):
        c = k[name]
        h = np.zeros(len(c["spheres"]), dtype=ob.HITTABLE_DTYPE)
        for i, s in enumerate(c["spheres"]):
            h[i]["centre"], h[i]["radius"], h[i]["texture"] = s["centre"], s["radius"], -1
            h[i]["albedo"] = [1, 1, 1]
        sc = ob.Scene(h)
        for mode in (ob.TRAVERSE_TREE, ob.TRAVERSE_GATED_BRUTE, ob.TRAVERSE_LINEAR):
            ids, t, _, _ = sc.trace([c["orig"]], [c["dir"]], mode)
            assert ids[0] == c["nearest_id"], (name, mode)
            if c.get("t_finite"):
                assert np.isfinite(t[0])


def test_seed1_scene_is_pinned(ob):
    g = golden("primary_rays_seed1.npz")
    h = ob.gen_scene(1)
    assert len(h) == 488
    assert hashlib.sha256(h.tobytes()).hexdigest() == str(g["scene_sha256"])


def test_golden_primary_rays_tree_equals_topology_free(ob):
    g = golden("primary_rays_seed1.npz")
    sc = ob.Scene(ob.gen_scene(1))
    assert sc.build_tree() == 975 and sc.tree_depth() == 10  # 2N-1 nodes (src/boundingvolume.rs:192-208)
    ids_b, t_b, n_b, ctr_b = sc.trace(g["orig"], g["dir"], ob.TRAVERSE_GATED_BRUTE)
    ids_t, t_t, n_t, ctr_t = sc.trace(g["orig"], g["dir"], ob.TRAVERSE_TREE)
    assert np.array_equal(ids_b, g["id"]) and np.array_equal(ids_t, g["id"])
    hit = g["id"] >= 0
    assert np.array_equal(t_b[hit], g["t"][hit]) and np.array_equal(t_t[hit], g["t"][hit])
    assert np.array_equal(n_b[hit], g["normal"][hit])
    # the reference tree visits a large share of its 975 boxes per ray (size-keyed split)
    assert ctr_t[0] / len(ids_t) > 40  # measured: about 95 of the 975 boxes per primary ray


def test_golden_render(ob):
    g = golden("render_seed1_48x32_8spp.npz")
    sc = ob.Scene(ob.gen_scene(1))
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 48, 32)
    s, q, st = sc.render(cam, spp=8, depth=10, seed=0, sampler=ob.SAMPLER_DIRECT)
    assert np.array_equal(s, g["sum_direct"])
    # thread count does not change any pixel; nor does the traversal flavour
    s1, _, _ = sc.render(cam, spp=8, depth=10, seed=0, sampler=ob.SAMPLER_DIRECT, threads=1, traverse=ob.TRAVERSE_TREE)
    assert np.array_equal(s1, s)
    s_ref, _, _ = sc.render(cam, spp=8, depth=10, seed=0, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    assert np.array_equal(s_ref, g["sum_reference"])
    # sample ranges add up: [0,8) = [0,3) + [3,8)
    a, _, _ = sc.render(cam, spp=3, depth=10, seed=0, sample_begin=0)
    b, _, _ = sc.render(cam, spp=5, depth=10, seed=0, sample_begin=3)
    assert np.allclose(a + b, s, rtol=1e-12, atol=0)
    # quantisation (src/lib.rs:186-192)
    assert np.array_equal(q, ob.finalize(s, 8))


def test_direct_and_reference_samplers_agree_statistically(ob):
    # the direct draws (unit sphere, quarter disc) must have the distributions of the reference's rejection loops
    sc = ob.Scene(ob.gen_scene(1))
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 32, 24)
    a, _, _ = sc.render(cam, spp=192, depth=10, seed=3, sampler=ob.SAMPLER_DIRECT)
    b, _, _ = sc.render(cam, spp=192, depth=10, seed=3, sampler=ob.SAMPLER_REFERENCE)
    c, _, _ = sc.render(cam, spp=192, depth=10, seed=4, sampler=ob.SAMPLER_REFERENCE)
    same = ob.psnr(b / 192, c / 192, peak=1.0)       # noise floor: same sampler, two seeds
    cross = ob.psnr(a / 192, b / 192, peak=1.0)
    assert abs(cross - same) < 1.0, (cross, same)
    assert abs(a.mean() - b.mean()) / b.mean() < 0.01


def test_texture_lookup_golden(ob, earth_rgb8):
    g = golden("earth_albedo_lookups.npz")
    h = np.zeros(1, dtype=ob.HITTABLE_DTYPE)
    h["material"], h["texture"], h["radius"] = 3, 0, 0.75
    h["centre"], h["orient_up"], h["orient_around"] = [0.0, 0.2, 3.0], [0.0, 1.0, -0.6], [1.0, 0.0, 0.5]
    sc = ob.Scene(h, [earth_rgb8])
    got = np.array([sc.albedo(0, x) for x in g["loc"]])
    assert np.array_equal(got, g["albedo"])
    # texel (0,0) of the decoded map is white; the north pole looks it up (latitude 0 -> row 0)
    up = np.array([0.0, 1.0, -0.6]) / np.linalg.norm([0.0, 1.0, -0.6])
    assert got.min() >= 0 and got.max() <= 1
    assert np.allclose(sc.albedo(0, up), earth_rgb8[0, sc_col(up)] / 255.0)


def sc_col(up):
    # longitude at the pole: atan2(a0.up, a1.up) + pi with a0 = around^, a1 = up^ x around^ (src/materials.rs:65-73)
    ar = np.array([1.0, 0.0, 0.5]) / np.linalg.norm([1.0, 0.0, 0.5])
    a1 = np.cross(up, ar)
    lon = np.arctan2(ar.dot(up), a1.dot(up)) + np.pi
    return int(0.5 * lon / np.pi * 1024) % 1024


def test_bundled_scene_json_renders_on_the_linear_path(ob, earth_rgb8):
    # config[0]: the bundled scene (disc + emitter + texture) through raytrace (src/lib.rs:40-86)
    import json as js

    with open(os.path.join(GOLDEN, "scene.json")) as f:
        doc = js.load(f)
    assert len(doc["hittables"]) == 5
    kinds = [list(x["shape"])[0] for x in doc["hittables"]]
    assert kinds == ["Sphere", "Sphere", "Disc", "Sphere", "Sphere"]
    h = np.zeros(5, dtype=ob.HITTABLE_DTYPE)
    mats = {"Diffuse": 0, "Metal": 1, "Dielectric": 2, "TextureMap": 3, "Emitter": 4}
    for i, e in enumerate(doc["hittables"]):
        (sk, s), = e["shape"].items()
        (mk, m), = e["material"].items()
        h[i]["shape"] = 0 if sk == "Sphere" else 1
        h[i]["centre"], h[i]["radius"], h[i]["texture"] = s["centre"], s["radius"], -1
        if sk == "Disc":
            h[i]["normal"] = s["normal"]
        h[i]["material"] = mats[mk]
        if "albedo" in m:
            h[i]["albedo"] = [m["albedo"]["r"], m["albedo"]["g"], m["albedo"]["b"]]
        h[i]["fuzz"] = m.get("fuzz", 0.0)
        h[i]["refractive_index"] = m.get("refractive_index", 0.0)
        if mk == "TextureMap":
            h[i]["texture"], h[i]["orient_up"], h[i]["orient_around"] = 0, m["orient_up"], m["orient_around"]
    sc = ob.Scene(h, [earth_rgb8])
    assert sc.build_tree() == -1  # the reference tree refuses a disc (unimplemented!, src/boundingvolume.rs:254-256)
    c = doc["camera"]
    cam = ob.camera_build(c["lookat"], c["lookfrom"], c["inv_focal_length"], c["aperture"], 40, 40)
    s, q, st = sc.render(cam, spp=4, depth=10, seed=0, traverse=ob.TRAVERSE_LINEAR)
    assert np.isfinite(s).all() and s.min() >= 0
    assert q.std() > 5  # an image, not a constant
