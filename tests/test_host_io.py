"""Host side of src/main.rs in the library: scene.json reader (src/config.rs + serde schema), the JPEG
decoder behind TextureMap's "map" (src/materials.rs:37-46), and the command-line front end.

No reference test covers these (parity unpinned against the reference's serde/image crates); the JPEG decoder
is pinned against libjpeg-turbo through PIL, which is a test-only dependency here."""
import io
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

PIL = pytest.importorskip("PIL.Image")

SCENE = json.load(open(os.path.join(GOLDEN, "scene.json")))
CLI = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")


def _jpeg(arr, **kw):
    b = io.BytesIO()
    PIL.fromarray(arr).save(b, "JPEG", **kw)
    raw = b.getvalue()
    return raw, np.asarray(PIL.open(io.BytesIO(raw)).convert("RGB"))


def _picture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 9.0), 128 + 100 * np.cos(y / 7.0), (x * 3 + y * 5) % 256], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


def test_jpeg_earthmap_is_byte_identical_to_libjpeg(rt, earth_rgb8):
    assert np.array_equal(rt.load_jpeg(os.path.join(GOLDEN, "earthmap.jpeg")), earth_rgb8)


@pytest.mark.parametrize("size", [(64, 64), (37, 51), (8, 8), (1, 1), (130, 17)])
@pytest.mark.parametrize("quality", [30, 75, 95])
def test_jpeg_444_matches_libjpeg_exactly(rt, size, quality):
    raw, want = _jpeg(_picture(*size), quality=quality, subsampling=0)
    assert np.array_equal(rt.decode_jpeg(raw), want)


def test_jpeg_grayscale_and_restart_intervals(rt):
    g = _picture(45, 70)[..., 0]
    raw, want = _jpeg(g, quality=80)
    assert np.array_equal(rt.decode_jpeg(raw), want)
    raw, want = _jpeg(_picture(64, 96), quality=85, subsampling=0, restart_marker_blocks=5)
    assert b"\xff\xd0" in raw
    assert np.array_equal(rt.decode_jpeg(raw), want)


@pytest.mark.parametrize("subsampling", [1, 2])
def test_jpeg_subsampled_chroma_is_close_to_libjpeg(rt, subsampling):
    # chroma is replicated here, libjpeg interpolates it: luma identical, colours close
    smooth = _picture(64, 80, seed=3)
    smooth[..., 2] = 90
    raw, want = _jpeg(smooth, quality=90, subsampling=subsampling)
    got = rt.decode_jpeg(raw).astype(int)
    assert got.shape == want.shape
    assert np.abs(got - want).mean() < 4.0


def test_jpeg_refuses_what_it_cannot_decode(rt):
    raw, _ = _jpeg(_picture(32, 32), quality=80, progressive=True)
    with pytest.raises(rt.RtError) as e:
        rt.decode_jpeg(raw)
    assert "progressive" in str(e.value)
    with pytest.raises(rt.RtError):
        rt.decode_jpeg(b"not a jpeg at all")
    ok, _ = _jpeg(_picture(32, 32), quality=80)
    with pytest.raises(rt.RtError):
        rt.decode_jpeg(ok[: len(ok) // 3])  # truncated before any image data is complete: no crash, an error or a partial image
    with pytest.raises(rt.RtError) as e:
        rt.load_jpeg("/nonexistent/file.jpeg")
    assert e.value.status == 4 and "cannot open file" in str(e.value)


def _text(doc):
    return json.dumps(doc)


def test_config_text_round_trip_and_unknown_keys(rt):
    doc = json.loads(json.dumps(SCENE))
    doc["comment"] = "ignored, as serde ignores unknown fields"
    doc["camera"]["extra"] = [1, 2, 3]
    doc["hittables"][0]["shape"]["Sphere"]["label"] = "ground é 🌍"
    cfg = rt.Config.from_text(_text(doc), texture_dir=GOLDEN)
    ref = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    assert cfg.hittables.tobytes() == ref.hittables.tobytes()
    assert bytes(cfg.camera.c) == bytes(ref.camera.c)
    want = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 512, 512)  # SetupCamera::setup (src/config.rs:24-27)
    assert bytes(cfg.camera.c) == bytes(want.c)


def test_config_numbers_follow_serde(rt):
    doc = json.loads(json.dumps(SCENE))
    doc["hittables"] = doc["hittables"][:1]
    doc["hittables"][0]["shape"]["Sphere"]["radius"] = 100  # an integer is a fine f64
    doc["hittables"][0]["shape"]["Sphere"]["centre"] = [1e-3, -2.5E+2, 0]
    cfg = rt.Config.from_text(_text(doc))
    assert cfg.hittables[0]["radius"] == 100.0 and list(cfg.hittables[0]["centre"]) == [0.001, -250.0, 0.0]
    doc["camera"]["horiz_res"] = 512.0  # a float is not a u32
    with pytest.raises(rt.RtError) as e:
        rt.Config.from_text(_text(doc).replace("512.0", "512.0"))
    assert "u32" in str(e.value) and "unable to deserialize scene information" in str(e.value)
    doc["camera"]["horiz_res"] = -1
    with pytest.raises(rt.RtError):
        rt.Config.from_text(_text(doc))


@pytest.mark.parametrize(
    "mutate, needle",
    [
        (lambda d: d["camera"].pop("aperture"), "missing field `aperture`"),
        (lambda d: d.pop("hittables"), "missing field `hittables`"),
        (lambda d: d["hittables"][0]["shape"].update({"Cube": {}}), "single key"),
        (lambda d: d["hittables"][0].update({"shape": {"Cube": {}}}), "unknown variant `Cube`"),
        (lambda d: d["hittables"][0].update({"material": {"Glow": {}}}), "unknown variant `Glow`"),
        (lambda d: d["hittables"][0]["material"]["Metal"]["albedo"].pop("g"), "missing field `g`"),
        (lambda d: d["hittables"][0]["shape"]["Sphere"].update({"centre": [0, 1]}), "invalid length 2"),
        (lambda d: d["hittables"][0]["shape"]["Sphere"].update({"radius": "big"}), "expected f64"),
        (lambda d: d["hittables"][3]["material"].update({"Dielectric": {}}), "missing field `refractive_index`"),
    ],
)
def test_config_errors_name_the_problem(rt, mutate, needle):
    doc = json.loads(json.dumps(SCENE))
    mutate(doc)
    with pytest.raises(rt.RtError) as e:
        rt.Config.from_text(_text(doc), texture_dir=GOLDEN)
    assert e.value.status == 4 and needle in str(e.value), str(e.value)


@pytest.mark.parametrize("text", ['{"camera": ', '{"camera": {}} trailing', "[1, 2,]", '{"a": 01}', '{"a": "\\x"}', ""])
def test_config_syntax_errors(rt, text):
    with pytest.raises(rt.RtError) as e:
        rt.Config.from_text(text)
    assert "unable to deserialize scene information" in str(e.value) and "line 1" in str(e.value)


def test_config_missing_files(rt, tmp_path):
    with pytest.raises(rt.RtError) as e:
        rt.Config.from_json(str(tmp_path / "nope.json"))
    assert "unable to read scene file" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.Config.from_text(_text(SCENE), texture_dir=str(tmp_path))  # earthmap.jpeg is not there
    assert "cannot open file" in str(e.value)


def test_cli_builds_and_prints_help():
    assert os.access(CLI, os.X_OK), "run __graft_entry__.build()"
    out = subprocess.run([CLI, "--help"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "--samples" in out.stdout and "--random" in out.stdout
    bad = subprocess.run([CLI, "--frobnicate"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


@pytest.mark.gpu
def test_cli_renders_scene_json_like_the_library(rt, tmp_path):
    out = tmp_path / "image.ppm"
    r = subprocess.run([CLI, "-s", "4", "--scene", os.path.join(GOLDEN, "scene.json"), "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Starting render..." in r.stdout and "Computing with %d samples" % (512 * 512 * 4) in r.stdout
    assert "Render finished in" in r.stdout
    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    _, want = rt.Scene(cfg.hittables, cfg.textures).render(cfg.camera, 4, depth=10, seed=0, want_sum=False)
    assert np.array_equal(_read_ppm(out), want.reshape(512, 512, 3))


@pytest.mark.gpu
def test_cli_random_scene(rt, tmp_path):
    out = tmp_path / "image.ppm"
    r = subprocess.run([CLI, "-r", "--samples", "3", "--width", "96", "--height", "64", "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "number of BoundingBox: 488" in r.stdout
    _, want = rt.Scene(rt.gen_scene(1)).render(rt.default_camera(96, 64), 3, depth=10, seed=0, want_sum=False)
    assert np.array_equal(_read_ppm(out), want.reshape(64, 96, 3))


@pytest.mark.gpu
def test_render_into_file_entry_points(rt, tmp_path):
    import ctypes as C

    from raytracer_amd import _abi

    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 40, 24)
    tex = (_abi.rt_texture * 1)(_abi.rt_texture(cfg.textures[0].ctypes.data, 1024, 512))
    path = tmp_path / "o.ppm"
    with open(path, "wb") as f:
        f.write(rt.ppm_header(cam).encode())
        f.flush()
        _abi.check(_abi.lib.rt_render_into_file(f.fileno(), C.byref(cam.c), cfg.hittables.ctypes.data, len(cfg.hittables), tex, 1,
                                                5, 10, 7, 0))
    _, want = rt.Scene(cfg.hittables, cfg.textures).render(cam, 5, depth=10, seed=7, want_sum=False)
    assert np.array_equal(_read_ppm(path), want.reshape(24, 40, 3))
    # the accelerated entry point covers spheres only, like the reference's make_covering
    with open(path, "wb") as f:
        rc = _abi.lib.rt_accel_render_into_file(f.fileno(), C.byref(cam.c), cfg.hittables.ctypes.data, len(cfg.hittables), tex, 1,
                                                5, 10, 7, 0)
    assert rc == 2 and b"not a sphere" in _abi.lib.rt_last_error()
    spheres = rt.gen_scene(1)
    with open(path, "wb") as f:
        f.write(rt.ppm_header(cam).encode())
        f.flush()
        _abi.check(_abi.lib.rt_accel_render_into_file(f.fileno(), C.byref(cam.c), spheres.ctypes.data, len(spheres), None, 0,
                                                      2, 10, 0, 0))
    _, want = rt.Scene(spheres).render(cam, 2, depth=10, seed=0, want_sum=False)
    assert np.array_equal(_read_ppm(path), want.reshape(24, 40, 3))
    # the binary variants append the same pixels as raw bytes behind a P6 header
    for fn, hs, tx, nt in ((_abi.lib.rt_accel_render_into_file_p6, spheres, None, 0), (_abi.lib.rt_render_into_file_p6, spheres, None, 0)):
        with open(path, "wb") as f:
            f.write(b"P6\n40 24\n255\n")
            f.flush()
            _abi.check(fn(f.fileno(), C.byref(cam.c), hs.ctypes.data, len(hs), tx, nt, 2, 10, 0, 0))
        assert path.read_bytes() == b"P6\n40 24\n255\n" + want.tobytes()


def test_checkpoint_file_round_trip(rt, tmp_path):
    import ctypes as C

    from raytracer_amd import _abi

    info = _abi.rt_checkpoint_info(7, 5, 12, 10, 99, 0xDEADBEEF)
    sums = np.arange(7 * 5 * 3, dtype=np.float32) * 0.25
    path = str(tmp_path / "a.rtacc").encode()
    _abi.check(_abi.lib.rt_checkpoint_save(path, C.byref(info), sums.ctypes.data))
    got = _abi.rt_checkpoint_info()
    _abi.check(_abi.lib.rt_checkpoint_load(path, C.byref(got), None, 0))  # header only
    assert (got.width, got.height, got.samples_done, got.depth, got.seed, got.tag) == (7, 5, 12, 10, 99, 0xDEADBEEF)
    back = np.zeros_like(sums)
    _abi.check(_abi.lib.rt_checkpoint_load(path, C.byref(got), back.ctypes.data, back.size))
    assert np.array_equal(back, sums)
    assert _abi.lib.rt_checkpoint_load(path, C.byref(got), back.ctypes.data, back.size - 1) == 1  # buffer too small
    open(tmp_path / "bad", "wb").write(b"P3\n1 1\n255\n")
    assert _abi.lib.rt_checkpoint_load(str(tmp_path / "bad").encode(), C.byref(got), None, 0) == 4
    raw = open(path, "rb").read()
    open(tmp_path / "short", "wb").write(raw[:-8])
    assert _abi.lib.rt_checkpoint_load(str(tmp_path / "short").encode(), C.byref(got), back.ctypes.data, back.size) == 4
    # the host finalize follows src/lib.rs:148-149, :186-192: mean, sqrt, x255, truncate, saturate
    q = np.zeros(4 * 3, dtype=np.uint8)
    s = np.array([4 * 1.0, 4 * 0.25, 0.0, 4 * 4.0, -1.0, np.nan, 4 * 0.999, 4 * 0.5, 1e-12, 0, 0, 0], dtype=np.float32)
    _abi.check(_abi.lib.rt_finalize(s.ctypes.data, 4, 4, q.ctypes.data))
    assert list(q[:9]) == [255, 127, 0, 255, 0, 0, 254, 180, 0]
    h = _abi.lib.rt_hash_bytes(b"abc", 3, 0)
    assert h == 0xE71FA2190541574B and _abi.lib.rt_hash_bytes(b"c", 1, _abi.lib.rt_hash_bytes(b"ab", 2, 0)) == h  # FNV-1a, chainable


@pytest.mark.gpu
def test_resumed_render_equals_uninterrupted(rt, tmp_path):
    from raytracer_amd import _abi

    scene = rt.Scene(rt.gen_scene(1))
    cam = rt.default_camera(96, 64)
    whole, q_whole = scene.render(cam, 12, depth=10, seed=5, samples_per_item=4)
    part, _ = scene.render(cam, 8, depth=10, seed=5, samples_per_item=4)
    o = scene.opts(8, 12, 10, 5, flags=_abi.RT_RENDER_ACCUMULATE, samples_per_item=4)
    q = np.zeros((64, 96, 3), dtype=np.uint8)
    _abi.check(_abi.lib.rt_render(scene.handle, cam.c, o, part.ctypes.data, q.ctypes.data))
    assert np.array_equal(part, whole) and np.array_equal(q.reshape(q_whole.shape), q_whole)
    q2 = np.zeros_like(q)
    _abi.check(_abi.lib.rt_finalize(whole.ctypes.data, 96 * 64, 12, q2.ctypes.data))
    assert np.array_equal(q2, q)  # host and device finalize agree byte for byte
    # the command line: two invocations with a checkpoint give the file of one
    args = [CLI, "-r", "--width", "96", "--height", "64", "--every", "4", "--seed", "5"]
    a, b, ck = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "b.rtacc"
    r = subprocess.run(args + ["-s", "12", "--out", str(a), "--checkpoint", str(tmp_path / "a.rtacc")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r1 = subprocess.run(args + ["-s", "8", "--out", str(b), "--checkpoint", str(ck)], capture_output=True, text=True, timeout=120)
    r2 = subprocess.run(args + ["-s", "12", "--out", str(b), "--checkpoint", str(ck)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert "Continuing from" in r2.stdout and "8 samples per pixel done" in r2.stdout
    assert open(a).read() == open(b).read()
    assert np.array_equal(_read_ppm(a), q_whole.reshape(64, 96, 3))
    other = subprocess.run(args[:-1] + ["6", "-s", "12", "--out", str(b), "--checkpoint", str(ck)], capture_output=True, text=True, timeout=120)
    assert other.returncode == 101 and "another scene, frame, depth or seed" in other.stderr


def test_jpeg_decoder_survives_damaged_files(rt):
    """Bit flips, truncation and garbage never crash the decoder: it returns an image or an error."""
    rng = np.random.default_rng(9)
    good, _ = _jpeg(_picture(48, 40), quality=85, subsampling=0, restart_marker_blocks=3)
    sub, _ = _jpeg(_picture(48, 40), quality=70, subsampling=2)
    outcomes = {"image": 0, "error": 0}
    for base in (good, sub):
        for trial in range(400):
            raw = bytearray(base)
            for _ in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(2, len(raw)))
                raw[pos] = int(rng.integers(0, 256))
            if trial % 5 == 0:
                raw = raw[: int(rng.integers(4, len(raw)))]
            try:
                img = rt.decode_jpeg(bytes(raw))
                assert img.ndim == 3 and img.shape[2] == 3 and img.size > 0
                outcomes["image"] += 1
            except rt.RtError as e:
                assert e.status in (2, 5)
                outcomes["error"] += 1
    assert outcomes["image"] > 50 and outcomes["error"] > 50, outcomes
    huge = bytearray(good)
    i = huge.index(b"\xff\xc0")
    huge[i + 5 : i + 9] = b"\xff\xff\xff\xff"  # 65535 x 65535
    with pytest.raises(rt.RtError):
        rt.decode_jpeg(bytes(huge))
