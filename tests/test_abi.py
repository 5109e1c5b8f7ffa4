"""CPU: the C-ABI library loads, exports what include/rt_mi355x.h declares, and its host-only entry points
(camera, scene generators, PPM writer, argument checks) behave like the reference / the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    from raytracer_amd import _abi

    names = _declared_functions()
    assert len(names) >= 18
    raw = C.CDLL(_abi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "library does not export %s" % n
    assert sorted(_abi.PROTOTYPES) == names, "ctypes prototypes and header disagree"
    assert _abi.lib.rt_abi_version() >> 16 == 1


def test_python_constants_match_the_header(rt):
    """Every RT_* enumerator of include/rt_mi355x.h that the ctypes mirror names has the header's value."""
    import os
    import re

    from raytracer_amd import _abi

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rt_mi355x.h")).read()
    found = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\b(RT_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))\b", text)}
    checked = [name for name in found if hasattr(_abi, name)]
    assert len(checked) >= 20 and "RT_RENDER_KERNEL_WAVEFRONT" in checked
    for name in checked:
        assert getattr(_abi, name) == found[name], name


def test_struct_sizes_match_oracle_layout(rt, ob):
    from raytracer_amd import _abi

    assert _abi.HITTABLE_DTYPE == ob.HITTABLE_DTYPE
    assert C.sizeof(_abi.rt_camera) == C.sizeof(ob.Camera) == 160


def test_camera_build_matches_oracle_bitwise(rt, ob):
    for args in [([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 1920, 1080), ([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 512, 512),
                 ([1, 2, 3], [-4, 0.5, 9], 0.7, 0.0, 33, 17)]:
        a = rt.Camera.build(*args)
        b = ob.camera_build(*args)
        assert bytes(a.c) == bytes(b)
    # src/camera.rs:27-33: up = +y, unit arms, aspect ratio computed (and unused)
    cam = rt.default_camera(1920, 1080)
    assert np.allclose(np.linalg.norm(cam.field("horiz_arm")), 1) and np.allclose(np.linalg.norm(cam.field("vert_arm")), 1)
    assert cam.field("aspect_ratio") == 1920 / 1080
    assert abs(cam.field("focal_distance") - np.sqrt(13 * 13 + 1.5 * 1.5 + 9)) < 1e-12


def test_camera_build_rejects_bad_arguments(rt):
    with pytest.raises(rt.RtError) as e:
        rt.Camera.build([0, 0, 0], [1, 1, 1], 1.0, 0.1, 0, 10)
    assert e.value.status == 1 and "zero resolution" in str(e.value)


def test_gen_scene_matches_oracle_bitwise(rt, ob):
    for seed in (1, 2, 12345):
        assert rt.gen_scene(seed).tobytes() == ob.gen_scene(seed).tobytes()
    s = rt.gen_scene(1)
    assert len(s) == 488  # ground + 3 + 22*22 (src/scenegen.rs:16-61)
    assert s[0]["radius"] == 1000 and list(s[0]["centre"]) == [0, -1000, 0]
    assert [int(m) for m in s["material"][:4]] == [0, 2, 0, 1]
    small = s[4:]
    assert np.all(small["radius"] == 0.2) and np.all(small["centre"][:, 1] == 0.2)
    cx = small["centre"][:, 0].reshape(22, 22)
    assert np.all((cx >= np.arange(-11, 11)[:, None]) & (cx < np.arange(-11, 11)[:, None] + 0.9))
    metal = small[small["material"] == 1]
    assert np.all((metal["albedo"] >= 0.5) & (metal["albedo"] < 1.0)) and np.all((metal["fuzz"] >= 0) & (metal["fuzz"] < 0.5))
    glass = small[small["material"] == 2]
    assert np.all((glass["refractive_index"] >= 1) & (glass["refractive_index"] < 2))


def test_stress_scene_generators(rt):
    g = rt.gen_scene_grid(3, 40)
    assert len(g) == 1601
    c = g["centre"][1:]
    # resting on the ground sphere, and no two small spheres overlap
    assert np.allclose(np.linalg.norm(c - np.array([0, -1000.0, 0]), axis=1), 1000.2)
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2) + np.eye(len(c)) * 10
    assert d.min() > 0.4
    sh = rt.gen_scene_shells(6)
    assert len(sh) == 8 and int(sh[1]["material"]) == 3 and int(sh[1]["texture"]) == 0
    ri = sh["refractive_index"][2:]
    assert np.allclose(ri[0::2], 1.5) and np.allclose(ri[1::2], 1 / 1.5)
    assert np.all(np.diff(sh["radius"][2:]) < 0)
    with pytest.raises(rt.RtError) as e:
        rt.gen_scene_grid(1, 0)
    assert e.value.status == 5


def test_write_ppm_format(rt, tmp_path):
    img = np.array([[[0, 1, 2], [255, 128, 7]], [[10, 20, 30], [99, 100, 101]]], dtype=np.uint8)
    p = tmp_path / "image.ppm"
    p.write_text("x" * 500)  # stale bytes must not survive (the reference leaves them, src/main.rs:35-39)
    rt.write_ppm(str(p), img)
    assert p.read_text() == "P3\n2 2\n255\n0 1 2\n255 128 7\n10 20 30\n99 100 101\n"
    with pytest.raises(rt.RtError) as e:
        rt.write_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), img)
    assert e.value.status == 4


def test_write_ppm_p6_byte_layout(rt, tmp_path):
    # binary PPM: "P6\nW H\n255\n" then W*H*3 raw bytes, rows top to bottom
    img = np.array([[[0, 1, 2], [255, 128, 7], [3, 4, 5]], [[10, 20, 30], [99, 100, 101], [6, 7, 8]]], dtype=np.uint8)
    p = tmp_path / "image.ppm"
    p.write_bytes(b"x" * 500)
    rt.write_ppm_p6(str(p), img)
    assert p.read_bytes() == b"P6\n3 2\n255\n" + img.tobytes()
    with pytest.raises(rt.RtError) as e:
        rt.write_ppm_p6(str(tmp_path / "no_such_dir" / "x.ppm"), img)
    assert e.value.status == 4


def test_scene_create_fails_loudly_without_a_gpu(rt):
    from raytracer_amd import _abi

    if _abi.lib.rt_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(rt.RtError) as e:
        rt.Scene(rt.gen_scene(1))
    assert e.value.status == 3  # RT_ERR_DEVICE: there is no CPU fallback


def test_config_from_json_reads_the_bundled_scene(rt, earth_rgb8):
    from conftest import GOLDEN

    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    assert cfg.camera.horiz_res == 512 and cfg.camera.vert_res == 512
    h = cfg.hittables
    assert [int(x) for x in h["shape"]] == [0, 0, 1, 0, 0]
    assert [int(x) for x in h["material"]] == [1, 4, 4, 2, 3]
    assert list(h[2]["normal"]) == [0.0, 1.0, 0.0] and h[2]["radius"] == 0.6
    assert h[0]["fuzz"] == 0.1 and h[3]["refractive_index"] == 1.5
    assert list(h[4]["orient_up"]) == [0.0, 1.0, -0.6] and int(h[4]["texture"]) == 0
    # "map" is a path resolved next to the scene file and decoded at load time (src/materials.rs:37-46)
    assert len(cfg.textures) == 1 and cfg.textures[0].shape == (512, 1024, 3)
    assert np.array_equal(cfg.textures[0], earth_rgb8)


def test_shapes_and_materials_helpers(rt):
    d = rt.disc([0, 0, 0], [0, 0, 2.0], 1.0, rt.emitter([1, 0.8, 1]))
    assert list(d["normal"]) == [0, 0, 1.0]  # Disc::new normalises (src/geometry.rs:129-131)
    d2 = rt.disc([0, 0, 0], [0, 0, 2.0], 1.0, rt.emitter([1, 0.8, 1]), normalise=False)
    assert list(d2["normal"]) == [0, 0, 2.0]  # serde does not
    s = rt.sphere([1, 2, 3], 2.0, rt.metal([0.8, 0.8, 0.8], 0.1))
    assert int(s["material"]) == 1 and s["fuzz"] == 0.1 and int(s["texture"]) == -1
    arr = rt.hittables([s, d])
    assert arr.dtype == rt.HITTABLE_DTYPE and len(arr) == 2


def test_scene_validation_needs_no_device(rt):
    """Bad scenes are refused before any device work, with the status that names the problem."""
    ok = rt.sphere([0, 0, 0], 1.0, rt.diffuse([.5, .5, .5]))
    for field, value, status in (("radius", float("nan"), 1), ("radius", float("inf"), 1), ("radius", 1e16, 1),
                                 ("shape", 7, 2), ("material", 9, 2)):
        h = rt.hittables([ok, ok])
        h[1][field] = value
        with pytest.raises(rt.RtError) as e:
            rt.Scene(h)
        assert e.value.status == status, (field, value, str(e.value))
    h = rt.hittables([ok])
    h[0]["centre"][1] = float("nan")
    with pytest.raises(rt.RtError) as e:
        rt.Scene(h)
    assert e.value.status == 1 and "hittable 0" in str(e.value)
    t = rt.hittables([rt.sphere([0, 0, 0], 1.0, rt.texture_map(3, [0, 1, 0], [1, 0, 0]))])
    with pytest.raises(rt.RtError) as e:
        rt.Scene(t)  # names texture 3 of 0
    assert e.value.status == 1
