"""GPU: the HIP path, called through the C ABI, against the CPU oracle and the committed golden vectors.

Bars: primitive IDs bit-exact; hit parameter and normal within 1e-5 relative (they come out at
f64 rounding level because the device runs the reference's f64 sphere test); per-path colours
within 2e-4 except the few paths where an f32 rounding flips a discrete decision; images by PSNR.
"""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

T_REL = 1e-5  # tolerance the parity target states for t and normals
# Largest difference, per component, between a ray direction the device forms (f32 arithmetic, one-instruction
# rcp / sqrt / sin / cos) and the oracle's f64 value of the same formula; test_generate_primary_matches_oracle
# holds the camera rays to it.
DIR_ERR = 5e-6


@pytest.fixture(scope="module")
def scene1(rt):
    return rt.Scene(rt.gen_scene(1))


@pytest.fixture(scope="module")
def oscene1(ob):
    return ob.Scene(ob.gen_scene(1))


def _ocam(ob, cam):
    return ob.camera_from_bytes(bytes(cam.c))


def _check_hits(ids, t, nrm, want_ids, want_t, want_n):
    assert np.array_equal(ids, want_ids), "primitive IDs differ at %s" % np.nonzero(ids != want_ids)[0][:10]
    hit = want_ids >= 0
    assert np.all(np.abs(t[hit] - want_t[hit]) <= T_REL * np.abs(want_t[hit]))
    assert np.all(np.abs(nrm[hit] - want_n[hit]) <= T_REL)
    return float(np.max(np.abs(t[hit] - want_t[hit]) / np.abs(want_t[hit]))) if hit.any() else 0.0


def test_scene_info(scene1):
    i = scene1.info
    assert i["n_hittables"] == 488 and i["n_spheres"] == 488 and i["n_discs"] == 0
    assert i["n_wide"] == 1  # the ground sphere covers all the others: tested per segment, outside the tree
    assert i["n_nodes"] == 486  # one sphere per leaf: N-1 inner nodes over the other 487
    assert 9 <= i["bvh_depth"] <= 16 and i["stack_depth"] == i["bvh_depth"] + 2


def test_primary_hits_match_golden_dump(scene1):
    g = golden("primary_rays_seed1.npz")
    ids, t, nrm = scene1.trace_primary(g["orig"], g["dir"])
    worst = _check_hits(ids, t, nrm, g["id"], g["t"], g["normal"])
    assert worst < 1e-12  # same f64 arithmetic: rounding level, far inside the 1e-5 bar


def test_primary_hits_random_rays_match_oracle(scene1, oscene1, ob):
    rng = np.random.default_rng(11)
    n = 60000
    o = np.concatenate([rng.uniform(-14, 14, (n // 2, 3)) * [1, 0.15, 1] + [0, 0.3, 0],   # among the spheres
                        rng.normal(size=(n // 2, 3)) * [20, 5, 20] + [0, 8, 0]]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d[: n // 4, 1] = -np.abs(d[: n // 4, 1])
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ids, t, nrm = scene1.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene1.trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert (want_ids >= 0).mean() > 0.3 and len(np.unique(want_ids)) > 300


def test_primary_hits_exact_zero_direction_components(scene1, oscene1, ob):
    # zero components take the 1e-4 divisor of the reference's box test (src/boundingvolume.rs:76-80)
    o, d = [], []
    for x in np.linspace(-10.7, 10.7, 41):
        o += [[x, 0.2, -30.0], [x, 30.0, 0.3], [-30.0, 0.2, x]]
        d += [[0, 0, 1.0], [0, -1.0, 0], [1.0, 0, 0]]
    o, d = np.array(o, dtype=np.float32), np.array(d, dtype=np.float32)
    ids, t, nrm = scene1.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene1.trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    tree_ids, _, _, _ = oscene1.trace(o, d, ob.TRAVERSE_TREE)
    assert np.array_equal(tree_ids, want_ids)


def test_generate_primary_matches_oracle(scene1, ob, rt):
    cam = rt.default_camera(96, 64)
    o, d = scene1.generate_primary(cam, seed=5, pixel_begin=100, pixel_end=1100, sample_begin=2, sample_end=5)
    wo, wd = ob.generate_primary(_ocam(ob, cam), 5, 100, 1100, 2, 5)
    assert np.abs(o - wo).max() < DIR_ERR and np.abs(d - wd).max() < DIR_ERR  # f32 camera arithmetic vs f64
    assert np.allclose(np.linalg.norm(d.astype(np.float64), axis=1), 1, atol=3e-7)  # Ray::new normalises
    # lens samples stay in the positive quarter disc (src/camera.rs:81-89)
    rel = (o.astype(np.float64) - cam.field("lookfrom")) / 0.1
    a, b = rel @ cam.field("horiz_arm"), rel @ cam.field("vert_arm")
    assert a.min() >= -1e-5 and b.min() >= -1e-5 and (a * a + b * b).max() <= 1 + 1e-5


def _path_parity(scene, oscene, cam, ocam, ob, n_pixels, spp, depth, seed, traverse, hittables):
    """Path-by-path comparison with the oracle run on the same Philox counters.

    Returns (got, want, close, same_ids, calm): colours, per-path closeness, equality of the primitive-ID
    sequences, and `calm` = paths that never enter a dielectric.  Where a ray refracts INTO a sphere the
    reference re-hits that sphere's own surface or not depending on the sign of f64 rounding noise
    (near root accepted for any t > 0, src/geometry.rs:115-118), so past such a vertex two correct
    implementations may legitimately part ways; everywhere else they must agree vertex by vertex.
    """
    rng = np.random.default_rng(3)
    pixel = np.repeat(rng.choice(cam.horiz_res * cam.vert_res, n_pixels, replace=False), spp).astype(np.uint32)
    sample = np.tile(np.arange(spp, dtype=np.uint32), n_pixels)
    got = scene.trace_paths(cam, pixel, sample, depth=depth, seed=seed).astype(np.float64)
    want = oscene.trace_paths(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, sampler=ob.SAMPLER_DIRECT)
    m = min(depth, 12) if depth else 1
    gi, gt, gn = scene.path_segments(cam, pixel, sample, depth=depth, seed=seed, max_segments=m)
    wi, wt, wn = oscene.path_segments(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, max_segments=m)
    close = np.all(np.abs(got - want) <= 2e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    same_ids = np.all(gi == wi, axis=1) & (gn == wn)
    mat = np.asarray(hittables["material"])
    calm = ~np.any((wi >= 0) & (mat[np.maximum(wi, 0)] == 2), axis=1)
    # wherever the sequences agree the parameters agree to the stated tolerance
    both = (gi == wi) & (wi >= 0) & same_ids[:, None]
    first = both[:, 0]
    assert np.all(np.abs(gt[first, 0] - wt[first, 0]) <= T_REL * np.abs(wt[first, 0]))  # f32 vs f64 camera arithmetic only
    # Later vertices drift: a bounce off a sphere of radius r magnifies a direction error by about (distance / r),
    # so no fixed tolerance holds along a whole path.  How much drift an error of DIR_ERR per direction causes on
    # THESE paths is measured on the oracle itself (the same paths traced again with every direction moved by up to
    # DIR_ERR); the device, whose direction error is below DIR_ERR, must agree with the oracle at least as often.
    def within(t, ids, n):
        ok = (ids == wi) & (wi >= 0) & (np.all(ids == wi, axis=1) & (n == wn))[:, None]
        return ok, np.abs(t - wt) <= 1e-4 * np.maximum(np.abs(wt), 0.1)
    oscene.set_dir_noise(DIR_ERR)
    try:
        ni, nt, nn = oscene.path_segments(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, max_segments=m)
        nwant = oscene.trace_paths(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, sampler=ob.SAMPLER_DIRECT)
    finally:
        oscene.set_dir_noise(0.0)
    ok_n, in_n = within(nt, ni, nn)
    ok_g, in_g = within(gt, gi, gn)
    expected = (ok_n & in_n).sum() / max((wi >= 0).sum(), 1)  # vertices that survive the stated direction error
    observed = (ok_g & in_g).sum() / max((wi >= 0).sum(), 1)
    assert observed >= expected - 0.005, (observed, expected)
    # the same yardstick for whole ID sequences: the share of glass-free paths whose sequence survives the stated
    # direction error; the callers hold the device to min(0.999, that share)
    noise_same = np.all(ni == wi, axis=1) & (nn == wn)
    _path_parity.id_floor = min(0.999, float(noise_same[calm].mean()) - 0.0005) if calm.any() else 0.999
    # and over ALL paths, glass included: past a refraction into a sphere the perturbed oracle parts from the
    # unperturbed one as any second correct implementation does, so its shares of equal ID sequences and of equal
    # colours are the bars for the device there too (binomial slack of three standard deviations)
    noise_close = np.all(np.abs(nwant - want) <= 2e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    slack = lambda share: 3.0 * np.sqrt(max(share * (1.0 - share), 1e-4) / len(pixel))
    _path_parity.all_id_floor = float(noise_same.mean()) - slack(float(noise_same.mean()))
    _path_parity.all_close_floor = float(noise_close.mean()) - slack(float(noise_close.mean()))
    return got, want, close, same_ids, calm


def test_paths_match_oracle_one_by_one(scene1, oscene1, ob, rt):
    cam = rt.default_camera(160, 90)
    got, want, close, same_ids, calm = _path_parity(scene1, oscene1, cam, _ocam(ob, cam), ob, 3000, 4, 10, 9,
                                                    ob.TRAVERSE_GATED_BRUTE, scene1.hittables)
    assert calm.mean() > 0.9
    assert same_ids[calm].mean() > 0.999, "ID sequences differ on %.5f of the paths without glass" % (1 - same_ids[calm].mean())
    assert close[calm].mean() > 0.998
    assert same_ids.mean() >= _path_parity.all_id_floor, (same_ids.mean(), _path_parity.all_id_floor)
    assert close.mean() >= _path_parity.all_close_floor, (close.mean(), _path_parity.all_close_floor)
    assert abs(got.mean() - want.mean()) < 5e-3 * want.mean()


def test_paths_depth_50(scene1, oscene1, ob, rt):
    cam = rt.default_camera(160, 90)
    got, want, close, same_ids, calm = _path_parity(scene1, oscene1, cam, _ocam(ob, cam), ob, 1500, 4, 50, 2,
                                                    ob.TRAVERSE_GATED_BRUTE, scene1.hittables)
    assert same_ids[calm].mean() > 0.999 and close[calm].mean() > 0.998
    assert abs(got.mean() - want.mean()) < 5e-3 * want.mean()


def test_depth_rules(scene1, rt):
    cam = rt.default_camera(32, 32)
    px = np.arange(64, dtype=np.uint32) * 7
    # depth 0: `for _ in 1..=0` never runs, the colour is the initial throughput (src/lib.rs:89-93, :134)
    assert np.all(scene1.trace_paths(cam, px, np.zeros(64, np.uint32), depth=0) == 1.0)
    s, _ = scene1.render(cam, 3, depth=0)
    assert np.all(s == 3.0)
    # depth 1: a hit returns the albedo (throughput after one scatter), a miss the sky
    one = scene1.trace_paths(cam, px, np.zeros(64, np.uint32), depth=1)
    assert one.min() >= 0 and one.max() <= 1.0


def test_render_matches_oracle_image(scene1, oscene1, ob, rt):
    cam = rt.default_camera(96, 64)
    s, q = scene1.render(cam, 16, depth=10, seed=0)
    want, want_q, st = oscene1.render(_ocam(ob, cam), spp=16, depth=10, seed=0, sampler=ob.SAMPLER_DIRECT)
    rel = np.abs(s - want) / np.maximum(want, 1.0)
    assert (rel.max(axis=2) < 1e-3).mean() > 0.95  # same counters, same draws: most pixels agree to f32 rounding
    assert ob.psnr(s / 16, want / 16, peak=1.0) > 30
    assert abs(s.sum() - want.sum()) < 1e-3 * want.sum()
    assert (np.abs(q.astype(int) - want_q.astype(int)) <= 1).mean() > 0.97
    # and against the reference's own sampling (rejection loops, reference tree): statistically
    ref, _, _ = oscene1.render(_ocam(ob, cam), spp=16, depth=10, seed=0, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    ref2, _, _ = oscene1.render(_ocam(ob, cam), spp=16, depth=10, seed=1, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    floor = ob.psnr(ref / 16, ref2 / 16, peak=1.0)
    assert abs(ob.psnr(s / 16, ref / 16, peak=1.0) - floor) < 1.0


def test_converged_image_psnr_against_reference_sampling(scene1, oscene1, ob, rt):
    # high spp, small image: device vs the oracle run with the reference's sampling and tree
    cam = rt.default_camera(64, 40)
    spp = 1024
    s, _ = scene1.render(cam, spp, depth=10, seed=21)
    a, _, _ = oscene1.render(_ocam(ob, cam), spp=spp, depth=10, seed=22, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    b, _, _ = oscene1.render(_ocam(ob, cam), spp=spp, depth=10, seed=23, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    floor = ob.psnr(a / spp, b / spp, peak=1.0)          # two independent reference-style renders
    ours = ob.psnr(s / spp, a / spp, peak=1.0)
    assert ours > floor - 0.5, (ours, floor)               # within 0.5 dB of the noise floor
    assert abs(s.mean() / spp - a.mean() / spp) < 0.003


def test_render_is_reproducible_and_ranges_add_up(scene1, rt):
    cam = rt.default_camera(100, 60)  # not a multiple of the 8x8 work tile
    a, qa = scene1.render(cam, 8, depth=10, seed=4)
    b, qb = scene1.render(cam, 8, depth=10, seed=4)
    assert np.array_equal(a, b) and np.array_equal(qa, qb)  # counter-based RNG, fixed summation order
    c, _ = scene1.render(cam, 8, depth=10, seed=5)
    assert not np.array_equal(a, c)
    lo, _ = scene1.render(cam, 3, depth=10, seed=4, sample_begin=0)
    hi, _ = scene1.render(cam, 5, depth=10, seed=4, sample_begin=3)
    assert np.allclose(lo + hi, a, rtol=2e-6, atol=1e-6)  # 2 "virtual ranks" = 1 rank, up to f32 summation order


def test_all_render_schedules_give_the_same_image(scene1, rt):
    # lock-step kernel, wave-scheduled kernel with the BVH in HBM (f32 boxes) and in LDS (f16 boxes):
    # the schedule and the box precision change which boxes are visited, never the hit that is found
    from raytracer_amd import _abi

    cam = rt.default_camera(200, 120)
    assert scene1.info["lds_resident"] == 1
    a, _ = scene1.render(cam, 6, depth=50, seed=8)
    b, _ = scene1.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_LOCKSTEP)
    c, _ = scene1.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_NO_LDS_SCENE)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    # the instrumented build runs the inner-node loop as compiled C++, the production build in assembly
    e, _ = scene1.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_STATS)
    assert np.array_equal(a, e)
    # and the image is the sum, in sample order, of the colours trace_paths returns path by path
    px = np.repeat(np.arange(200 * 120, dtype=np.uint32), 6)
    sm = np.tile(np.arange(6, dtype=np.uint32), 200 * 120)
    cols = scene1.trace_paths(cam, px, sm, depth=50, seed=8).reshape(200 * 120, 6, 3)

    def ordered_sum(chunk):  # chunks of samples summed in sample order, then added in chunk order
        total = np.zeros((200 * 120, 3), dtype=np.float32)
        for c0 in range(0, 6, chunk):
            part = np.zeros((200 * 120, 3), dtype=np.float32)
            for k in range(c0, min(c0 + chunk, 6)):
                part += cols[:, k, :]
            total += part
        return total.reshape(120, 200, 3)

    assert np.array_equal(ordered_sum(scene1.render_chunk(cam, 6)), a)
    for chunk in (1, 4, 6):
        d, _ = scene1.render(cam, 6, depth=50, seed=8, samples_per_item=chunk)
        assert scene1.render_chunk(cam, 6, samples_per_item=chunk) == chunk
        assert np.array_equal(ordered_sum(chunk), d)
    assert np.allclose(a, ordered_sum(6), rtol=2e-6, atol=1e-6)  # the chunking only reorders an f32 sum


def test_wavefront_renderer_gives_the_same_image(scene1, rt, earth_rgb8, monkeypatch):
    """RT_RENDER_KERNEL_WAVEFRONT: generate / trace / shade kernels that hand rays on through queues in device memory,
    compacted by the material they hit.  Every path is the persistent kernel's path (same counters, same arithmetic)
    and a pixel's samples are added in the same order, so the two images must be equal bit for bit: tree in LDS, tree
    in HBM, no tree (nested shells), discs + emitter + texture map; one batch and many; accumulation on top."""
    import os

    from conftest import GOLDEN
    from raytracer_amd import _abi

    wf = _abi.RT_RENDER_KERNEL_WAVEFRONT
    cam = rt.default_camera(200, 120)
    a, _ = scene1.render(cam, 6, depth=50, seed=8)
    b, _ = scene1.render(cam, 6, depth=50, seed=8, flags=wf)
    assert np.array_equal(a, b)
    for chunk in (1, 4):
        c, _ = scene1.render(cam, 6, depth=50, seed=8, samples_per_item=chunk)
        d, _ = scene1.render(cam, 6, depth=50, seed=8, samples_per_item=chunk, flags=wf)
        assert np.array_equal(c, d)
    # depth 0 (every path returns white) and depth 1
    for depth in (0, 1):
        c, _ = scene1.render(cam, 3, depth=depth, seed=1)
        d, _ = scene1.render(cam, 3, depth=depth, seed=1, flags=wf)
        assert np.array_equal(c, d)
    # a pool of 6 * 1000 slots: 24 batches of 1000 pixels
    monkeypatch.setenv("RT_WF_SLOTS", "6000")
    fresh = rt.Scene(rt.gen_scene(1))
    e, _ = fresh.render(cam, 6, depth=50, seed=8, flags=wf)
    assert np.array_equal(a, e)
    monkeypatch.delenv("RT_WF_SLOTS")
    grid = rt.Scene(rt.gen_scene_grid(2, 48))  # tree in HBM
    assert grid.info["lds_resident"] == 0
    gcam = rt.Camera.build([0, 0, 0], [8, 3, 2], 1.0, 0.05, 96, 64)
    f, _ = grid.render(gcam, 4, depth=50, seed=3)
    g, _ = grid.render(gcam, 4, depth=50, seed=3, flags=wf)
    assert np.array_equal(f, g)
    shells = rt.Scene(rt.gen_scene_shells(6), [earth_rgb8])  # no tree: every sphere is tested per segment
    scam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 96, 96)
    h, _ = shells.render(scam, 4, depth=50, seed=2)
    i, _ = shells.render(scam, 4, depth=50, seed=2, flags=wf)
    assert np.array_equal(h, i)
    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))  # disc, emitter, texture map
    bundled = rt.Scene(cfg.hittables, cfg.textures)
    j, _ = bundled.render(scam, 4, depth=10, seed=0)
    k, _ = bundled.render(scam, 4, depth=10, seed=0, flags=wf)
    assert np.array_equal(j, k)
    # no hittable at all, and a lone sphere
    for hs in (rt.hittables([]), rt.hittables([rt.sphere([0, 0.2, 3.0], 0.75, rt.metal([0.8, 0.6, 0.2], 0.3))])):
        tiny = rt.Scene(hs)
        tcam = rt.Camera.build([0, 0.2, 3.0], [0, 1.5, -5], 1.0, 0.0, 40, 24)
        m, _ = tiny.render(tcam, 3, depth=10, seed=5)
        n, _ = tiny.render(tcam, 3, depth=10, seed=5, flags=wf)
        assert np.array_equal(m, n)
    # samples [0, 3) and then [3, 6) on top, on the device, one chunk and several: the sums meet the same way
    import ctypes as C

    import hip_mem

    for chunk in (0, 2):
        sums = []
        for flags in (0, wf):
            d = hip_mem.DeviceArray(120 * 200 * 3)
            for (s0, s1, acc) in ((0, 3, 0), (3, 6, _abi.RT_RENDER_ACCUMULATE)):
                o = scene1.opts(s0, s1, 50, 8, flags | acc, chunk)
                assert _abi.lib.rt_render_device(scene1.handle, C.byref(cam.c), C.byref(o), d.ptr, None) == 0, _abi.lib.rt_last_error()
            sums.append(d.numpy().reshape(120, 200, 3))
            d.free()
        assert np.array_equal(sums[0], sums[1])
        whole, _ = scene1.render(cam, 6, depth=50, seed=8, samples_per_item=chunk)
        assert np.allclose(sums[1], whole, rtol=2e-6, atol=1e-6)
    # the entry point that allocates nothing refuses it: the path pool is memory of the scene's own
    d = hip_mem.DeviceArray(120 * 200 * 3)
    o = scene1.opts(0, 2, 5, 0, wf, 0)
    rc = _abi.lib.rt_render_device_noalloc(scene1.handle, C.byref(cam.c), C.byref(o), d.ptr, None)
    d.free()
    assert rc == _abi.RT_ERR_INVALID_ARGUMENT


def test_work_hand_out_covers_every_pixel_sample_once(scene1, rt):
    """Eight work counters, tiles that hang over the frame's edge, frames smaller than a tile, waves that find their
    counter dry and move on, bands (tree in HBM) and interleaved items (scene in LDS): whatever the frame, the image is
    the chunk-ordered sum of the per-path colours, so every (pixel, sample) was rendered exactly once."""
    grid = rt.Scene(rt.gen_scene_grid(2, 48))  # tree in HBM: chunks of two samples, one band of the frame per counter
    assert grid.info["lds_resident"] == 0 and scene1.info["lds_resident"] == 1
    for scene in (scene1, grid):
        for (w, h, spp) in ((1, 1, 3), (7, 3, 5), (9, 17, 2), (64, 8, 1), (33, 41, 7), (8, 8, 64)):
            cam = rt.default_camera(w, h)
            img, _ = scene.render(cam, spp, depth=8, seed=11, want_rgb8=False)
            px = np.repeat(np.arange(w * h, dtype=np.uint32), spp)
            sm = np.tile(np.arange(spp, dtype=np.uint32), w * h)
            cols = scene.trace_paths(cam, px, sm, depth=8, seed=11).reshape(w * h, spp, 3)
            chunk = scene.render_chunk(cam, spp)
            total = np.zeros((w * h, 3), dtype=np.float32)
            for c0 in range(0, spp, chunk):
                part = np.zeros((w * h, 3), dtype=np.float32)
                for k in range(c0, min(c0 + chunk, spp)):
                    part += cols[:, k, :]
                total += part
            assert np.array_equal(total.reshape(h, w, 3), img), (w, h, spp, chunk)
    assert grid.render_chunk(rt.default_camera(64, 64), 64) == 2 and grid.render_chunk(rt.default_camera(64, 64), 1000) == 16


def test_finalize_matches_color_to_ppm(scene1, ob, rt):
    cam = rt.default_camera(64, 64)
    s, q = scene1.render(cam, 4, depth=10, seed=1)
    assert np.array_equal(q, ob.finalize(s.astype(np.float64), 4))  # (255*sqrt(c)) as u8 on the same sums


def test_empty_and_single_sphere_scenes(rt, ob):
    cam = rt.default_camera(16, 16)
    ocam = _ocam(ob, cam)
    empty = rt.Scene(rt.hittables([]))
    s, _ = empty.render(cam, 2, depth=10, seed=0)
    w, _, _ = ob.Scene(np.zeros(0, dtype=ob.HITTABLE_DTYPE)).render(ocam, spp=2, depth=10, seed=0, traverse=ob.TRAVERSE_LINEAR)
    assert np.allclose(s, w, rtol=1e-5)  # sky only
    ids, _, _ = empty.trace_primary([[0, 0, 0]], [[0, 0, 1]])
    assert ids[0] == -1
    one_h = rt.hittables([rt.sphere([0, 0, 0], 5.0, rt.diffuse([0.4, 0.2, 0.1]))])  # src/scenegen.rs:111-127
    one = rt.Scene(one_h)
    ids, t, n = one.trace_primary([[10, 0, 0], [10, 0, 0], [0, 0, 0]], [[1, 0, 0], [-1, 0, 0], [0, 1, 0]])
    assert list(ids) == [-1, 0, 0] and t[1] == 5.0 and t[2] == 5.0 and list(n[1]) == [1, 0, 0]


def test_reference_inline_cases_on_device(rt):
    # src/geometry.rs:162-174 and src/boundingvolume.rs:451-502 through the device path
    s = rt.Scene(rt.hittables([rt.sphere([0, 0, 0], 2.0, rt.diffuse([1, 1, 1]))]))
    ids, t, n = s.trace_primary([[0, 0, -3]], [[0, 0, 1]])
    assert ids[0] == 0 and t[0] == 1.0 and list(n[0]) == [0, 0, -1]
    s = rt.Scene(rt.hittables([rt.sphere([0, 0, 0], 1.0, rt.diffuse([1, 1, 1]))]))
    assert s.trace_primary([[2, 0, 0]], [[1, 0, 0]])[0][0] == -1
    three = rt.Scene(rt.hittables([rt.sphere([0, 0, 0], 5.0, rt.diffuse([1, 1, 1])), rt.sphere([0, 0, 2], 1.0, rt.diffuse([1, 1, 1])),
                                   rt.sphere([0, 2, 0], 1.0, rt.diffuse([1, 1, 1]))]))
    ids, t, _ = three.trace_primary([[-1.5, -0.5, -0.5]], [[1, 0, 0]])
    assert ids[0] == 0 and np.isfinite(t[0]) and t[0] > 6


def test_equal_parameters_resolve_to_lowest_id(rt):
    dup = rt.hittables([rt.sphere([0, 0, 5], 1.0, rt.diffuse([1, 0, 0])), rt.sphere([3, 0, 5], 1.0, rt.diffuse([0, 1, 0])),
                        rt.sphere([0, 0, 5], 1.0, rt.diffuse([0, 0, 1]))])
    s = rt.Scene(dup)
    ids, _, _ = s.trace_primary([[0, 0, 0], [3, 0, 0]], [[0, 0, 1], [0, 0, 1]])
    assert list(ids) == [0, 1]


def test_bundled_scene_json_disc_emitter_texture(rt, ob, earth_rgb8):
    import os

    from conftest import GOLDEN

    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    scene = rt.Scene(cfg.hittables, cfg.textures)
    assert scene.info["n_discs"] == 1 and scene.info["n_spheres"] == 4
    oscene = ob.Scene(cfg.hittables, cfg.textures)
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 128, 128)
    ocam = _ocam(ob, cam)
    # nearest hits incl. the disc's negative parameters (src/geometry.rs:133-139 has no h > 0 test)
    o, d = scene.generate_primary(cam, seed=0, pixel_begin=0, pixel_end=128 * 128)
    ids, t, nrm = scene.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene.trace(o, d, ob.TRAVERSE_LINEAR)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert (want_ids == 2).any()
    # a disc BEHIND the origin wins with its negative parameter
    ids, t, nrm = scene.trace_primary([[0.7, 1.0, 0.7], [0.9, 3.0, 0.5]], [[0, 1, 0], [0, 1, 0]])
    assert list(ids) == [2, 2] and list(t) == [-1.25, -3.25] and list(nrm[0]) == [0, 1, 0]
    got, want, close, same_ids, calm = _path_parity(scene, oscene, cam, ocam, ob, 1500, 4, 10, 6, ob.TRAVERSE_LINEAR, cfg.hittables)
    assert same_ids[calm].mean() > 0.999 and close[calm].mean() > 0.995
    assert abs(got.mean() - want.mean()) < 1e-2 * want.mean()
    s, q = scene.render(cam, 8, depth=10, seed=0)
    w, wq, _ = oscene.render(ocam, spp=8, depth=10, seed=0, traverse=ob.TRAVERSE_LINEAR)
    assert ob.psnr(s / 8, w / 8, peak=1.0) > 30


def test_accel_render_refuses_discs_like_the_reference_tree(rt, earth_rgb8, tmp_path):
    import os

    from conftest import GOLDEN

    cfg = rt.Config.from_json(os.path.join(GOLDEN, "scene.json"))
    scene = rt.Scene(cfg.hittables, cfg.textures)
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 16, 16)
    with open(tmp_path / "a.ppm", "w") as f:
        with pytest.raises(rt.RtError):
            rt.accel_render_into_file(f, cam, scene, 1)  # unimplemented!() at src/boundingvolume.rs:254-256
    with open(tmp_path / "image.ppm", "w") as f:
        f.write(rt.ppm_header(cam))
        rt.render_into_file(f, cam, scene, 2)
    lines = open(tmp_path / "image.ppm").read().split("\n")
    assert lines[:3] == ["P3", "16 16", "255"] and len(lines) == 3 + 256 + 1
    assert all(0 <= int(v) <= 255 for v in lines[3].split())


def test_texture_albedo_on_device(rt, ob, earth_rgb8):
    # a lone textured sphere lit by the sky: depth-1 colours are the texel colours
    h = rt.hittables([rt.sphere([0, 0.2, 3.0], 0.75, rt.texture_map(0, [0, 1, -0.6], [1, 0, 0.5]))])
    scene, oscene = rt.Scene(h, [earth_rgb8]), ob.Scene(h, [earth_rgb8])
    cam = rt.Camera.build([0, 0.2, 3.0], [0, 1.5, -5], 0.2, 0.0, 96, 96)
    px = np.arange(96 * 96, dtype=np.uint32)
    got = scene.trace_paths(cam, px, np.zeros_like(px), depth=1, seed=0).astype(np.float64)
    want = oscene.trace_paths(_ocam(ob, cam), px, np.zeros_like(px), depth=1, seed=0, traverse=ob.TRAVERSE_LINEAR)
    same = np.all(np.abs(got - want) < 1e-6, axis=1)
    assert same.mean() > 0.995  # f32 acos/atan2 can cross a texel boundary


def test_nested_dielectric_shells_deep_bounces(rt, ob, earth_rgb8):
    h = rt.gen_scene_shells(6)
    scene, oscene = rt.Scene(h, [earth_rgb8]), ob.Scene(h, [earth_rgb8])
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 96, 96)
    got, want, close, same_ids, calm = _path_parity(scene, oscene, cam, _ocam(ob, cam), ob, 2500, 4, 50, 1,
                                                    ob.TRAVERSE_GATED_BRUTE, h)
    assert same_ids[calm].mean() > 0.999 and close[calm].mean() > 0.998
    assert (~calm).mean() > 0.1  # the shells are actually being looked at
    # through the shells the two implementations part ways vertex-wise but not statistically
    assert abs(got[~calm].mean() - want[~calm].mean()) < 0.03 * want[~calm].mean()
    assert abs(got.mean() - want.mean()) < 1e-2 * want.mean()


def _render_paths_against_oracle(scene, oscene, cam, ocam, ob, spp, depth, seed, traverse, hittables):
    """Per-path colours from the RENDER kernel itself (production code built with one extra store per path)
    against the oracle on the same counters.  Returns (got, want, close, calm) over all W*H*spp paths."""
    n = cam.horiz_res * cam.vert_res
    got = scene.render_paths(cam, spp, depth=depth, seed=seed).reshape(n * spp, 3).astype(np.float64)
    pixel = np.repeat(np.arange(n, dtype=np.uint32), spp)
    sample = np.tile(np.arange(spp, dtype=np.uint32), n)
    want = oscene.trace_paths(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, sampler=ob.SAMPLER_DIRECT)
    wi, _, _ = oscene.path_segments(ocam, pixel, sample, depth=depth, seed=seed, traverse=traverse, max_segments=min(depth, 12))
    mat = np.asarray(hittables["material"])
    calm = ~np.any((wi >= 0) & (mat[np.maximum(wi, 0)] == 2), axis=1)
    close = np.all(np.abs(got - want) <= 2e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    # the dump is the image: summed in the render's chunk order it gives the render's sums bit for bit
    img, _ = scene.render(cam, spp, depth=depth, seed=seed)
    chunk = scene.render_chunk(cam, spp)
    cols = got.reshape(n, spp, 3).astype(np.float32)
    total = np.zeros((n, 3), dtype=np.float32)
    for c0 in range(0, spp, chunk):
        part = np.zeros((n, 3), dtype=np.float32)
        for k in range(c0, min(c0 + chunk, spp)):
            part += cols[:, k, :]
        total += part
    assert np.array_equal(total.reshape(cam.vert_res, cam.horiz_res, 3), img)
    # and it is what the per-path entry point returns (same arithmetic, different kernel)
    assert np.array_equal(got.astype(np.float32), scene.trace_paths(cam, pixel, sample, depth=depth, seed=seed))
    return got, want, close, calm


def test_tree_scene_with_disc_emitter_and_texture_runs_the_full_build(rt, ob, earth_rgb8):
    """The production kernel exists twice: a lean build for scenes of plain spheres and the full one.  The bench
    scene plus a disc, an emitter and a textured sphere is a scene WITH a tree that needs the full build: nearest hits
    against the oracle's scan, the image against the ordered sum of its own per-path dump and against the wavefront
    renderer, and the plain scene's two builds (RT_NO_PLAIN in the environment of a second scene) against one another."""
    import os

    from raytracer_amd import _abi

    base = rt.gen_scene(1)
    extra = rt.hittables([
        rt.sphere([6.0, 0.6, 2.5], 0.6, rt.texture_map(0, [0, 1, -0.6], [1, 0, 0.5])),
        rt.sphere([5.0, 2.2, 0.5], 0.4, rt.emitter([4.0, 3.5, 3.0])),
        rt.disc([7.0, 0.6, 0.5], [0, 1, 0], 0.8, rt.metal([0.9, 0.9, 0.9], 0.05)),
    ])
    h = np.concatenate([base, extra])
    scene, oscene = rt.Scene(h, [earth_rgb8]), ob.Scene(h, [earth_rgb8])
    assert scene.info["n_nodes"] > 0 and scene.info["n_discs"] == 1
    cam = rt.default_camera(160, 96)
    o, d = scene.generate_primary(cam, seed=0, pixel_begin=0, pixel_end=160 * 96)
    ids, t, nrm = scene.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene.trace(o, d, ob.TRAVERSE_LINEAR)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    for added in range(len(base), len(base) + 3):  # the three additions are in the picture
        assert (ids == added).any()
    spp = 4
    img, _ = scene.render(cam, spp, depth=20, seed=3)
    cols = scene.render_paths(cam, spp, depth=20, seed=3).reshape(160 * 96, spp, 3)
    chunk = scene.render_chunk(cam, spp)
    total = np.zeros((160 * 96, 3), dtype=np.float32)
    for c0 in range(0, spp, chunk):
        part = np.zeros((160 * 96, 3), dtype=np.float32)
        for k in range(c0, min(c0 + chunk, spp)):
            part += cols[:, k, :]
        total += part
    assert np.array_equal(total.reshape(96, 160, 3), img)
    wf, _ = scene.render(cam, spp, depth=20, seed=3, flags=_abi.RT_RENDER_KERNEL_WAVEFRONT)
    assert np.array_equal(wf, img)
    # without the disc the scene runs the build that leaves the disc code out: same identity
    nodisc = rt.Scene(h[:-1], [earth_rgb8])
    assert nodisc.info["n_nodes"] > 0 and nodisc.info["n_discs"] == 0
    img2, _ = nodisc.render(cam, spp, depth=20, seed=3)
    cols2 = nodisc.render_paths(cam, spp, depth=20, seed=3).reshape(160 * 96, spp, 3)
    chunk2 = nodisc.render_chunk(cam, spp)
    total2 = np.zeros((160 * 96, 3), dtype=np.float32)
    for c0 in range(0, spp, chunk2):
        part = np.zeros((160 * 96, 3), dtype=np.float32)
        for k in range(c0, min(c0 + chunk2, spp)):
            part += cols2[:, k, :]
        total2 += part
    assert np.array_equal(total2.reshape(96, 160, 3), img2)
    # the same for a tree that is read from memory (grid of spheres + an emitter: the no-disc build of that kernel)
    g = np.concatenate([rt.gen_scene_grid(2, 48), rt.hittables([rt.sphere([3.0, 4.0, 1.0], 1.5, rt.emitter([3.0, 3.0, 2.5]))])])
    grid = rt.Scene(g)
    assert grid.info["lds_resident"] == 0 and grid.info["n_discs"] == 0
    gcam = rt.Camera.build([0, 0, 0], [8, 3, 2], 1.0, 0.05, 96, 64)
    img3, _ = grid.render(gcam, spp, depth=20, seed=4)
    cols3 = grid.render_paths(gcam, spp, depth=20, seed=4).reshape(96 * 64, spp, 3)
    chunk3 = grid.render_chunk(gcam, spp)
    total3 = np.zeros((96 * 64, 3), dtype=np.float32)
    for c0 in range(0, spp, chunk3):
        part = np.zeros((96 * 64, 3), dtype=np.float32)
        for k in range(c0, min(c0 + chunk3, spp)):
            part += cols3[:, k, :]
        total3 += part
    assert np.array_equal(total3.reshape(64, 96, 3), img3)
    # the plain scene through both builds of its kernel
    lean, _ = rt.Scene(base).render(cam, spp, depth=20, seed=3)
    os.environ["RT_NO_PLAIN"] = "1"
    try:
        full, _ = rt.Scene(base).render(cam, spp, depth=20, seed=3)
    finally:
        del os.environ["RT_NO_PLAIN"]
    assert np.array_equal(lean, full)


def test_render_kernel_paths_lds_scene_against_oracle(scene1, oscene1, ob, rt):
    # the kernel of the bench workload: scene in LDS, f16 boxes, assembly inner loop
    assert scene1.info["lds_resident"] == 1
    cam = rt.default_camera(96, 54)
    got, want, close, calm = _render_paths_against_oracle(scene1, oscene1, cam, _ocam(ob, cam), ob, 3, 50, 5,
                                                          ob.TRAVERSE_GATED_BRUTE, scene1.hittables)
    assert calm.mean() > 0.9 and close[calm].mean() > 0.998
    assert abs(got.mean() - want.mean()) < 5e-3 * want.mean()


def test_render_kernel_paths_hbm_tree_against_oracle(rt, ob):
    # 2305 spheres: too large for LDS, so the render kernel walks the quantised tree in HBM (inner_loop_hbm)
    h = rt.gen_scene_grid(2, 48)
    scene, oscene = rt.Scene(h), ob.Scene(h)
    assert scene.info["lds_resident"] == 0
    cam = rt.Camera.build([0, 0, 0], [20, 6, 8], 1.0, 0.05, 96, 60)
    ocam = _ocam(ob, cam)
    got, want, close, calm = _render_paths_against_oracle(scene, oscene, cam, ocam, ob, 2, 50, 3, ob.TRAVERSE_GATED_BRUTE, h)
    assert calm.mean() > 0.85 and close[calm].mean() > 0.998
    assert abs(got.mean() - want.mean()) < 5e-3 * want.mean()
    # vertex by vertex on sampled paths (per-path entry point, which the dump above equals bit for bit)
    g2, w2, close2, same_ids, calm2 = _path_parity(scene, oscene, cam, ocam, ob, 1500, 2, 50, 3, ob.TRAVERSE_GATED_BRUTE, h)
    # dense field, depth 50: many short hops between spheres, so more ID sequences part ways under the stated
    # direction error than in the sparse scene; the oracle's own perturbed run sets the bar
    assert same_ids[calm2].mean() > _path_parity.id_floor, (same_ids[calm2].mean(), _path_parity.id_floor)
    assert close2[calm2].mean() > _path_parity.id_floor - 0.001


def test_render_kernel_paths_shells_against_oracle(rt, ob, earth_rgb8):
    h = rt.gen_scene_shells(8)
    scene, oscene = rt.Scene(h, [earth_rgb8]), ob.Scene(h, [earth_rgb8])
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 64, 64)
    got, want, close, calm = _render_paths_against_oracle(scene, oscene, cam, _ocam(ob, cam), ob, 2, 50, 1, ob.TRAVERSE_GATED_BRUTE, h)
    assert close[calm].mean() > 0.998 and (~calm).mean() > 0.1
    assert abs(got[~calm].mean() - want[~calm].mean()) < 0.03 * want[~calm].mean()


def test_shells_converged_image_psnr_against_reference_sampling(rt, ob, earth_rgb8):
    """Earth map + nested dielectric shells, converged: the device against two independent oracle renders made with
    the reference's own sampling (rejection loops, sequential stream) and its tree.  The near-root rule of
    src/geometry.rs:115-120 (a refracted ray may re-hit the surface it starts on) shapes this image."""
    h = rt.gen_scene_shells(8)
    scene, oscene = rt.Scene(h, [earth_rgb8]), ob.Scene(h, [earth_rgb8])
    cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 64, 40)
    ocam = _ocam(ob, cam)
    spp = 1024
    s, _ = scene.render(cam, spp, depth=50, seed=31)
    a, _, _ = oscene.render(ocam, spp=spp, depth=50, seed=32, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    b, _, _ = oscene.render(ocam, spp=spp, depth=50, seed=33, sampler=ob.SAMPLER_REFERENCE, traverse=ob.TRAVERSE_TREE)
    floor = ob.psnr(a / spp, b / spp, peak=1.0)
    ours = 0.5 * (ob.psnr(s / spp, a / spp, peak=1.0) + ob.psnr(s / spp, b / spp, peak=1.0))
    assert ours > floor - 0.5, (ours, floor)
    assert abs(s.mean() - a.mean()) / spp < 0.003


def test_negative_radius_sphere_is_a_hollow_shell(rt, ob):
    """A sphere with a negative radius has the same surface and an inward normal (the hollow-glass idiom).  The
    reference's linear path handles it (the radius enters Sphere::intersect squared, src/geometry.rs:108); here it
    also sits in the tree, whose boxes are built from |r|."""
    rng = np.random.default_rng(4)
    grey = rt.diffuse([.5, .5, .5])
    items = [rt.sphere([0, 0, 0], 1.0, rt.dielectric(1.5)), rt.sphere([0, 0, 0], -0.9, rt.dielectric(1.5)),
             rt.sphere([2.5, 0, 0], -0.7, grey), rt.sphere([0, -101, 0], 100.0, grey)]
    items += [rt.sphere(list(rng.uniform(-4, 4, 3)), -0.3 if i % 2 else 0.3, grey) for i in range(40)]
    h = rt.hittables(items)
    o = rng.uniform(-5, 5, (8000, 3)).astype(np.float32)
    d = rng.normal(size=(8000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_LINEAR)
    for builder in ("sah", "lbvh"):
        scene = rt.Scene(h, builder=builder)
        ids, t, nrm = scene.trace_primary(o, d)
        _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert (h["radius"][np.maximum(want_ids, 0)][want_ids >= 0] < 0).sum() > 150  # the inside-out spheres are being hit
    cam = rt.Camera.build([0, 0, 0], [0, 1, -5], 1.0, 0.05, 64, 48)
    scene, oscene = rt.Scene(h), ob.Scene(h)
    got, want, close, calm = _render_paths_against_oracle(scene, oscene, cam, _ocam(ob, cam), ob, 2, 20, 2, ob.TRAVERSE_LINEAR, h)
    assert close[calm].mean() > 0.998


def test_far_origins_on_a_wide_sphere(rt, ob):
    """Secondary rays start on the wide spheres, far outside the box of the tree's spheres: here a mirror dome of
    radius 200 around a cluster of radius 1 sends nearly every camera ray back through the cluster.  The origin
    widening of the f32 descent has to cover the rounding of such origins (it is sized from the wide spheres too)."""
    rng = np.random.default_rng(6)
    items = [rt.sphere([0, 0, 0], 200.0, rt.metal([0.9, 0.9, 0.9], 0.0))]
    for _ in range(300):
        c = rng.normal(size=3); c = c / np.linalg.norm(c) * rng.uniform(0.2, 1.0)
        items.append(rt.sphere(list(c), 0.03, rt.diffuse(list(rng.uniform(0.2, 0.9, 3)))))
    h = rt.hittables(items)
    scene, oscene = rt.Scene(h), ob.Scene(h)
    assert scene.info["n_wide"] == 1
    # rays from the dome's surface aimed to graze the small spheres: the parity entry point sizes the widening
    # from the origins it is given
    n = 20000
    o = rng.normal(size=(n, 3)); o = (o / np.linalg.norm(o, axis=1, keepdims=True) * 199.9).astype(np.float32)
    k = rng.integers(1, len(items), n)
    aim = h["centre"][k] + rng.normal(0, 0.025, (n, 3))
    d = aim - o.astype(np.float64); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ids, t, nrm = scene.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene.trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert (want_ids > 0).mean() > 0.2
    # the render kernels size it from the scene (wide spheres included) and the camera
    cam = rt.Camera.build([0, 0, 0.6], [0.05, 0.02, -0.1], 1.5, 0.0, 64, 48)  # inside the cluster: the dome sends the rays back here
    n_px = 64 * 48
    got = scene.render_paths(cam, 4, depth=6, seed=1).reshape(n_px * 4, 3).astype(np.float64)
    pixel = np.repeat(np.arange(n_px, dtype=np.uint32), 4)
    sample = np.tile(np.arange(4, dtype=np.uint32), n_px)
    want = oscene.trace_paths(_ocam(ob, cam), pixel, sample, depth=6, seed=1, traverse=ob.TRAVERSE_GATED_BRUTE)
    wi, _, _ = oscene.path_segments(_ocam(ob, cam), pixel, sample, depth=6, seed=1, traverse=ob.TRAVERSE_GATED_BRUTE, max_segments=6)
    gi, _, _ = scene.path_segments(cam, pixel, sample, depth=6, seed=1, max_segments=6)
    assert ((wi[:, 1:] > 0).any(axis=1)).mean() > 0.05  # rays coming back from the dome do hit the cluster
    # up to the first hit after the dome the two sides differ by the f32 rounding of one mirror direction only
    assert np.all(gi[:, :2] == wi[:, :2], axis=1).mean() > 0.999
    assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(1.0, np.abs(want)), axis=1).mean() > 0.99


def test_grid_scene_primary_hits(rt, ob):
    h = rt.gen_scene_grid(2, 48)  # 2305 spheres on the ground sphere
    scene, oscene = rt.Scene(h), ob.Scene(h)
    cam = rt.Camera.build([0, 0, 0], [20, 6, 8], 1.0, 0.05, 128, 80)
    o, d = scene.generate_primary(cam, seed=1, pixel_begin=0, pixel_end=128 * 80)
    ids, t, nrm = scene.trace_primary(o, d)
    want_ids, want_t, want_n, _ = oscene.trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert len(np.unique(want_ids)) > 500


def test_full_size_properties_1080p(scene1, rt):
    # the bench workload's size: properties that need no oracle
    cam = rt.default_camera(1920, 1080)
    s, q = scene1.render(cam, 2, depth=50, seed=0)
    assert s.shape == (1080, 1920, 3) and np.isfinite(s).all() and s.min() >= 0 and s.max() <= 2.0 + 1e-5
    assert 0.4 < s.mean() / 2 < 0.9
    a, _ = scene1.render(cam, 1, depth=50, seed=0, sample_begin=0)
    b, _ = scene1.render(cam, 1, depth=50, seed=0, sample_begin=1)
    assert np.allclose(a + b, s, rtol=2e-6, atol=1e-6)
    s2, q2 = scene1.render(cam, 2, depth=50, seed=0)
    assert np.array_equal(q, q2)


def test_million_sphere_scene_primary_hits_and_render(rt, ob):
    # the 1M-sphere stress configuration at full size: BVH in HBM (does not fit LDS), f32 boxes
    h = rt.gen_scene_grid(1, 1000)
    assert len(h) == 1000001
    scene = rt.Scene(h)
    info = scene.info
    assert info["n_wide"] == 1 and info["n_nodes"] == 999999 and info["lds_resident"] == 0 and info["bvh_depth"] <= 27
    cam = rt.Camera.build([0, 0, 0], [130, 40, 30], 1.0, 0.1, 1920, 1080)
    rng = np.random.default_rng(1)
    px = rng.choice(1920 * 1080, 384, replace=False)
    o = np.concatenate([scene.generate_primary(cam, 0, int(p), int(p) + 1)[0] for p in px])
    d = np.concatenate([scene.generate_primary(cam, 0, int(p), int(p) + 1)[1] for p in px])
    # add secondary-like rays: from points just above the field in random upward directions
    o2 = (rng.uniform(-200, 200, (128, 3)) * [1, 0, 1] + [0, 0.45, 0]).astype(np.float32)
    o2[:, 1] = (np.sqrt(1000.2 ** 2 - o2[:, 0].astype(np.float64) ** 2 - o2[:, 2].astype(np.float64) ** 2) - 1000.0 + 0.25).astype(np.float32)
    d2 = rng.normal(size=(128, 3)); d2[:, 1] = np.abs(d2[:, 1]) * 0.05
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    ids, t, nrm = scene.trace_primary(o, d)
    want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    assert (want_ids > 0).sum() > 150  # small spheres, not only the ground
    # 2048 whole paths from the render kernel (tree in HBM, quantised nodes, assembly loop) against the oracle's
    # scan over all million spheres; the bar is the share of glass-free paths that survives the stated direction
    # error in the oracle itself
    small = rt.Camera.build([0, 0, 0], [130, 40, 30], 1.0, 0.1, 64, 32)
    oscene = ob.Scene(h)
    n_paths = 64 * 32
    got = scene.render_paths(small, 1, depth=50, seed=7).reshape(n_paths, 3).astype(np.float64)
    pxs = np.arange(n_paths, dtype=np.uint32)
    zeros = np.zeros(n_paths, np.uint32)
    want = oscene.trace_paths(_ocam(ob, small), pxs, zeros, depth=50, seed=7, traverse=ob.TRAVERSE_GATED_BRUTE)
    wi, _, _ = oscene.path_segments(_ocam(ob, small), pxs, zeros, depth=50, seed=7, traverse=ob.TRAVERSE_GATED_BRUTE, max_segments=12)
    oscene.set_dir_noise(DIR_ERR)
    try:
        nwant = oscene.trace_paths(_ocam(ob, small), pxs, zeros, depth=50, seed=7, traverse=ob.TRAVERSE_GATED_BRUTE)
    finally:
        oscene.set_dir_noise(0.0)
    calm = ~np.any((wi >= 0) & (np.asarray(h["material"])[np.maximum(wi, 0)] == 2), axis=1)
    tol = 2e-4 * np.maximum(1.0, np.abs(want))
    close = np.all(np.abs(got - want) <= tol, axis=1)
    noise_close = np.all(np.abs(nwant - want) <= tol, axis=1)
    floor = min(0.998, float(noise_close[calm].mean()) - 0.001)
    assert calm.sum() > 1500 and close[calm].mean() >= floor, (calm.sum(), close[calm].mean(), floor)
    assert np.array_equal(got.astype(np.float32), scene.trace_paths(small, pxs, zeros, depth=50, seed=7))
    assert (wi[:, 0] > 0).sum() > 60  # the camera rays do reach the small spheres
    # full-size render: finite, reproducible, sample ranges add up, schedules agree
    from raytracer_amd import _abi
    a, _ = scene.render(cam, 2, depth=50, seed=0, want_rgb8=False)
    b, _ = scene.render(cam, 2, depth=50, seed=0, want_rgb8=False, flags=_abi.RT_RENDER_KERNEL_LOCKSTEP)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    lo, _ = scene.render(cam, 1, depth=50, seed=0, sample_begin=0, want_rgb8=False)
    hi, _ = scene.render(cam, 1, depth=50, seed=0, sample_begin=1, want_rgb8=False)
    assert np.allclose(lo + hi, a, rtol=2e-6, atol=1e-6)


def test_4k_frame_properties(scene1, rt):
    # the 8-GPU configuration's frame size (3840x2160) at a token sample count
    cam = rt.default_camera(3840, 2160)
    s, q = scene1.render(cam, 1, depth=50, seed=0)
    assert s.shape == (2160, 3840, 3) and np.isfinite(s).all() and 0.4 < s.mean() < 0.9
    # same field of view as the 960x540 frame (the resolution only sets the pixel grid, src/camera.rs:59-78):
    # 4x4 blocks of the 4K frame at 1 spp estimate the pixels of the small frame at 16 spp
    small, _ = scene1.render(rt.default_camera(960, 540), 16, depth=50, seed=1)
    blocks = s.reshape(540, 4, 960, 4, 3).sum(axis=(1, 3))
    assert abs(blocks.mean() - small.mean()) < 0.01 * small.mean()
    coarse_a = blocks.reshape(54, 10, 96, 10, 3).mean(axis=(1, 3))
    coarse_b = small.reshape(54, 10, 96, 10, 3).mean(axis=(1, 3))
    assert np.corrcoef(coarse_a.ravel(), coarse_b.ravel())[0, 1] > 0.99


def test_argument_errors(scene1, rt):
    cam = rt.default_camera(16, 16)
    with pytest.raises(rt.RtError) as e:
        scene1.render(cam, 0)
    assert e.value.status == 1 and "empty sample range" in str(e.value)
    with pytest.raises(rt.RtError):
        scene1.trace_paths(cam, np.array([16 * 16], dtype=np.uint32), np.array([0], dtype=np.uint32))
    with pytest.raises(rt.RtError):
        scene1.trace_primary([[np.nan, 0, 0]], [[0, 0, 1]])
    bad = rt.gen_scene(1)[:3].copy()
    bad[1]["material"] = 9
    with pytest.raises(rt.RtError) as e:
        rt.Scene(bad)
    assert e.value.status == 2


def test_lbvh_builder_same_hits_same_image(rt, ob, scene1):
    """The tree built on the GPU (Morton order) is another conservative tree: every nearest hit, hence every
    path and the whole image, is identical to the host-built SAH tree's."""
    h = rt.gen_scene(1)
    lb = rt.Scene(h, builder="lbvh")
    info = lb.info
    assert info["n_wide"] == 1 and info["n_nodes"] == 486 and info["lds_resident"] == 1
    assert 9 <= info["bvh_depth"] <= 40 and info["stack_depth"] == info["bvh_depth"] + 2
    g = golden("primary_rays_seed1.npz")
    ids, t, nrm = lb.trace_primary(g["orig"], g["dir"])
    assert np.array_equal(ids, g["id"])
    ids0, t0, nrm0 = scene1.trace_primary(g["orig"], g["dir"])
    assert np.array_equal(t, t0) and np.array_equal(nrm, nrm0)
    cam_small = rt.default_camera(160, 96)
    a, qa = lb.render(cam_small, 6, depth=50, seed=3)
    b, qb = scene1.render(cam_small, 6, depth=50, seed=3)
    assert np.array_equal(a, b) and np.array_equal(qa, qb)


def test_lbvh_builder_degenerate_inputs(rt, ob):
    rng = np.random.default_rng(5)
    cases = {
        "two": [rt.sphere([0, 0, 0], 1.0, rt.diffuse([.5, .5, .5])), rt.sphere([3, 0, 0], 1.0, rt.diffuse([.5, .5, .5]))],
        "three_in_a_row": [rt.sphere([float(i) * 2.5, 0, 0], 1.0, rt.diffuse([.5, .5, .5])) for i in range(3)],
        # equal centres give equal Morton codes: the position in the sorted order splits them
        "coincident": [rt.sphere([1, 2, 3], 0.5 + 0.1 * i, rt.diffuse([.5, .5, .5])) for i in range(9)],
        "clustered": [rt.sphere(list(rng.normal(0, 1e-4, 3) + (i % 3) * 5.0), 0.3, rt.diffuse([.5, .5, .5])) for i in range(40)],
    }
    for name, items in cases.items():
        h = rt.hittables(items)
        o = rng.uniform(-8, 12, (4000, 3)).astype(np.float32)
        aim = h["centre"][rng.integers(0, len(h), 4000)] + rng.normal(0, 0.4, (4000, 3))  # towards some sphere, roughly
        d = aim - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        ids, t, nrm = rt.Scene(h, builder="lbvh").trace_primary(o, d)
        want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_GATED_BRUTE)
        assert (want_ids >= 0).sum() > 500, name
        _check_hits(ids, t, nrm, want_ids, want_t, want_n)


def test_lbvh_builder_million_spheres(rt):
    h = rt.gen_scene_grid(1, 1000)
    sah, lb = rt.Scene(h), rt.Scene(h, builder="lbvh")
    assert lb.info["n_nodes"] == 999999 and lb.info["n_wide"] == 1 and lb.info["bvh_depth"] <= 64
    assert lb.info["build_seconds"] < sah.info["build_seconds"]
    cam = rt.Camera.build([0, 0, 0], [130, 40, 30], 1.0, 0.1, 1920, 1080)
    rng = np.random.default_rng(2)
    px = rng.choice(1920 * 1080, 4096, replace=False)
    o, d = zip(*[lb.generate_primary(cam, 0, int(p), int(p) + 1) for p in px[:512]])
    o, d = np.concatenate(o), np.concatenate(d)
    a, b = lb.trace_primary(o, d), sah.trace_primary(o, d)
    assert (a[0] > 0).sum() > 100
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ra, _ = lb.render(cam, 1, depth=50, seed=0, want_rgb8=False)
    rb, _ = sah.render(cam, 1, depth=50, seed=0, want_rgb8=False)
    assert np.array_equal(ra, rb)


def test_lbvh_tree_that_stays_on_the_device(rt, ob):
    """From 4096 spheres up the GPU-built tree never visits the host: nodes renumbered depth-first, quantised copy and
    sphere tables made by device kernels.  Hits against the oracle's scan over all spheres, image against the host-built tree's."""
    from raytracer_amd import _abi
    h = rt.gen_scene_grid(3, 70)  # 4 901 spheres
    lb, sah = rt.Scene(h, builder="lbvh"), rt.Scene(h)
    assert lb.info["n_nodes"] == len(h) - 2 and lb.info["n_wide"] == 1 and lb.info["lds_resident"] == 0
    cam = rt.Camera.build([0, 0, 0], [20, 6, 5], 1.0, 0.1, 192, 108)
    o, d = lb.generate_primary(cam, 0, 0, 192 * 108)
    ids, t, nrm = lb.trace_primary(o, d)
    want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    assert (want_ids > 0).sum() > 2000
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)
    for flags in (0, _abi.RT_RENDER_KERNEL_LOCKSTEP):  # Node32 made on the device / the f32 nodes as renumbered
        a, _ = lb.render(cam, 4, depth=50, seed=2, want_rgb8=False, flags=flags)
        b, _ = sah.render(cam, 4, depth=50, seed=2, want_rgb8=False, flags=flags)
        assert np.array_equal(a, b)


def test_random_scenes_across_size_thresholds(rt, ob):
    """Random scenes of any scale and position (a tenth of the radii negative, sometimes a giant sphere underneath)
    on both sides of every rule that depends on the sphere count: no tree (5 / 6), scene in LDS or not, tree kept
    on the device (4095 / 4096).  Both builders: nearest hits against the oracle's scan over all spheres, and the
    render kernel's image against the ordered sum of the per-path entry point's colours."""
    rng = np.random.default_rng(7)
    for n in (1, 2, 5, 6, 12, 13, 300, 520, 1500, 4095, 4096, 6000):  # 300: five-wave build of the LDS kernel, 520: four-wave
        for giant in (False, True):
            scale = 10.0 ** rng.uniform(-2, 3)
            shift = rng.normal(0, 1, 3) * scale * 10.0 ** rng.uniform(-1, 2)
            items = [rt.sphere(list(rng.uniform(-1, 1, 3) * scale + shift), float(scale * 10.0 ** rng.uniform(-3, -0.5) * (1 if rng.random() > 0.1 else -1)),
                               rt.diffuse([.5, .5, .5])) for _ in range(n)]
            if giant and n > 1:
                items[0] = rt.sphere([float(shift[0]), float(shift[1] - 1001 * scale), float(shift[2])], 1000.0 * scale, rt.diffuse([.5, .5, .5]))
            h = rt.hittables(items)
            m = 2000
            o = (rng.uniform(-2, 2, (m, 3)) * scale + shift).astype(np.float32)
            aim = h["centre"][rng.integers(0, n, m)] + rng.normal(0, 0.3, (m, 3)) * np.abs(h["radius"][rng.integers(0, n, m)])[:, None]
            d = aim - o
            d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
            want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_GATED_BRUTE)
            cam = rt.Camera.build([float(x) for x in shift], [float(x) for x in (shift + np.array([2.5, 1.0, 2.0]) * scale)], 1.0, 0.05 * scale, 40, 24)
            px = np.repeat(np.arange(40 * 24, dtype=np.uint32), 3)
            sm = np.tile(np.arange(3, dtype=np.uint32), 40 * 24)
            for builder in ("sah", "lbvh"):
                sc = rt.Scene(h, builder=builder)
                ids, t, nrm = sc.trace_primary(o, d)
                assert np.array_equal(ids, want_ids), (n, giant, builder)
                hit = want_ids >= 0
                assert np.allclose(t[hit], want_t[hit], rtol=1e-9, atol=0), (n, giant, builder)
                img, _ = sc.render(cam, 3, depth=6, seed=1, want_rgb8=False, samples_per_item=3)
                cols = sc.trace_paths(cam, px, sm, depth=6, seed=1).reshape(40 * 24, 3, 3)
                total = np.zeros((40 * 24, 3), dtype=np.float32)
                for k in range(3):
                    total += cols[:, k, :]
                assert np.array_equal(total.reshape(24, 40, 3), img), (n, giant, builder)


def test_render_kernel_keeps_its_occupancy(rt, scene1):
    # one register too many (129 instead of 128) once cut a quarter of the resident waves and cost 40 % of the speed
    import ctypes as C

    from raytracer_amd import _abi

    b, w, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _abi.check(_abi.lib.rt_debug_render_launch(scene1.handle, C.byref(b), C.byref(w), C.byref(s)))
    assert (b.value, w.value, s.value) == (256, 5, 5)  # the five-wave build: only the nodes in LDS, five workgroups to a CU
    big = rt.Scene(rt.gen_scene_grid(1, 60))  # 3601 spheres: too large for LDS
    assert big.info["lds_resident"] == 0
    _abi.check(_abi.lib.rt_debug_render_launch(big.handle, C.byref(b), C.byref(w), C.byref(s)))
    assert b.value == 256 and s.value == 5  # shallow tree: five workgroups' stacks fit LDS, the five-wave build runs
    # a scene without a tree runs the shade-only kernel, which needs fewer registers: five waves per SIMD
    grey = rt.diffuse([.5, .5, .5])
    few = rt.Scene(rt.hittables([rt.sphere([i, 0, 0], 0.4, grey) for i in range(5)]))
    assert few.info["n_nodes"] == 0
    _abi.check(_abi.lib.rt_debug_render_launch(few.handle, C.byref(b), C.byref(w), C.byref(s)))
    assert b.value == 256 and w.value == s.value and s.value >= 5


def test_four_and_five_wave_builds_render_the_same_image(rt, scene1):
    """The kernel of a tree in LDS exists in two builds: five waves per SIMD with only the nodes in LDS (used when five
    workgroups fit a CU), and four waves with spheres and material tables in LDS as well.  RT_FIVE_WAVES=0 when the
    scene is created picks the second one; with the same samples per work item both must give the same bits."""
    import ctypes as C
    import os

    from raytracer_amd import _abi

    os.environ["RT_FIVE_WAVES"] = "0"
    try:
        four = rt.Scene(rt.gen_scene(1))
    finally:
        del os.environ["RT_FIVE_WAVES"]
    b, w, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _abi.check(_abi.lib.rt_debug_render_launch(four.handle, C.byref(b), C.byref(w), C.byref(s)))
    assert (b.value, w.value, s.value) == (512, 2, 4)
    _abi.check(_abi.lib.rt_debug_render_launch(scene1.handle, C.byref(b), C.byref(w), C.byref(s)))
    assert (b.value, w.value, s.value) == (256, 5, 5)
    cam = rt.default_camera(160, 96)
    for chunk in (3, 7):
        a, qa = four.render(cam, 7, depth=50, seed=11, samples_per_item=chunk)
        c, qc = scene1.render(cam, 7, depth=50, seed=11, samples_per_item=chunk)
        assert np.array_equal(a, c) and np.array_equal(qa, qc), chunk
    # left to themselves the two builds cut a pixel's samples into different chunks
    assert four.render_chunk(cam, 100) >= scene1.render_chunk(cam, 100)


def test_wide_spheres_outside_the_tree(rt, ob):
    """Spheres whose box covers most of the others are tested per segment instead of through the tree; which
    spheres those are must not change any hit."""
    rng = np.random.default_rng(11)
    grey = rt.diffuse([.5, .5, .5])
    small = [rt.sphere(list(rng.uniform(-3, 3, 3)), 0.3, grey) for _ in range(30)]
    cases = {
        # name: (hittables, wide spheres expected)
        # up to five spheres there is no tree at all: every sphere is tested per segment
        "big_and_small": ([rt.sphere([0, 0, 0], 50.0, grey), rt.sphere([1, 0, 0], 0.5, grey)], 2),
        "concentric_pair": ([rt.sphere([0, 0, 0], 2.0, grey), rt.sphere([0, 0, 0], 1.0, grey)], 2),
        "five": (small[:5], 5),
        "six": (small[:6], 0),
        "a_dozen": (small[:12], 0),
        # up to a dozen spheres around a group of nested shells: no tree either
        "shells_and_a_few": ([rt.sphere([0, 0, 0], 1.0 + 0.5 * i, rt.dielectric(1.5)) for i in range(5)] + small[:7], 12),
        "shells_and_too_many": ([rt.sphere([0, 0, 0], 1.0 + 0.5 * i, rt.dielectric(1.5)) for i in range(5)] + small[:8], 3),
        "three_shells_and_crowd": ([rt.sphere([0, 0, 0], 40.0, grey), rt.sphere([0, 0, 0], 30.0, rt.dielectric(1.5)),
                                    rt.sphere([0.5, 0, 0], 20.0, rt.dielectric(1.5))] + small, 3),
        "many_shells": ([rt.sphere([0, 0, 0], 10.0 + i, rt.dielectric(1.5)) for i in range(7)] + small, 4),  # at most four
        "no_wide": (small, 0),
    }
    for name, (items, n_wide) in cases.items():
        h = rt.hittables(items)
        scene = rt.Scene(h)
        assert scene.info["n_wide"] == n_wide, name
        assert scene.info["n_spheres"] == len(items)
        o = rng.uniform(-6, 6, (6000, 3)).astype(np.float32)
        d = rng.normal(size=(6000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        ids, t, nrm = scene.trace_primary(o, d)
        want_ids, want_t, want_n, _ = ob.Scene(h).trace(o, d, ob.TRAVERSE_GATED_BRUTE)
        _check_hits(ids, t, nrm, want_ids, want_t, want_n)
        assert len(np.unique(want_ids)) >= min(len(items), 3), name
        # the render kernels (wide spheres in trav_begin) against the per-path entry point (wide spheres in closest_hit)
        cam = rt.Camera.build([0, 0, 0], [0, 1, -5], 1.0, 0.1, 64, 48)
        img, _ = scene.render(cam, 3, depth=20, seed=2)
        px = np.repeat(np.arange(64 * 48, dtype=np.uint32), 3)
        sm = np.tile(np.arange(3, dtype=np.uint32), 64 * 48)
        cols = scene.trace_paths(cam, px, sm, depth=20, seed=2).reshape(64 * 48, 3, 3)
        total = np.zeros((64 * 48, 3), dtype=np.float32)
        for k in range(3):
            total += cols[:, k, :]
        assert np.array_equal(total.reshape(48, 64, 3), img), name


def test_nested_shell_rule_edge_cases(rt):
    """Of a group of concentric, well separated spheres the render kernels test only the three a segment can meet
    first (chosen per lane), and every shell when an origin of the wave is far away.  The per-path entry point tests
    every sphere of the scene, so image == ordered sum of its colours holds only if the rule never drops the nearest hit."""
    glass, grey = rt.dielectric(1.5), rt.diffuse([.6, .5, .4])
    lamp = rt.sphere([0, 30, 0], 8.0, rt.emitter([4, 4, 4]))
    near = ([0, 0, 0], [0, 1.5, -6], 1.0)
    cases = {
        # name: (hittables, (lookat, lookfrom, inverse focal length), shells in the group)
        "tight": ([rt.sphere([0, 0, 0], 1.0 * (1.0 + 1.2e-4) ** i, glass) for i in range(6)] + [lamp], near, 6),
        "small_and_barely_apart_enough": ([rt.sphere([0, 0, 0], 0.03 + 5.0e-6 * i, glass) for i in range(5)] + [lamp],
                                          ([0, 0, 0], [0, 0.05, -0.2], 1.0), 5),
        "small_and_not_apart_enough": ([rt.sphere([0, 0, 0], 0.03 + 3.5e-6 * i, glass) for i in range(5)] + [lamp],
                                       ([0, 0, 0], [0, 0.05, -0.2], 1.0), 0),
        "too_close_to_be_a_group": ([rt.sphere([0, 0, 0], 1.0 + 2.0e-5 * i, glass) for i in range(5)] + [lamp], near, 0),
        "mixed_signs": ([rt.sphere([0.3, 0.2, 0], r, glass) for r in (1.5, -1.4, 1.2, -1.1, 0.8, -0.7, 0.4)] + [lamp], near, 7),
        "ids_interleaved": ([rt.sphere([0, 0, 0], 2.0, glass), rt.sphere([2.5, 0, 0], 0.4, grey), rt.sphere([0, 0, 0], 1.0, glass),
                             lamp, rt.sphere([0, 0, 0], 1.5, glass), rt.sphere([0, 0, 0], 0.5, glass), rt.sphere([0, 0, 0], 1.75, glass)], near, 5),
        # origins farther than a thousand smallest radii: the wave tests every shell (camera rays), then the
        # scattered rays, which start on the shells, take the short way again
        "far_camera": ([rt.sphere([0, 0, 0], 0.5 + 0.1 * i, glass) for i in range(6)] + [lamp], ([0, 0, 0], [0, 300, -1500], 0.002), 6),
    }
    for name, (items, (lookat, lookfrom, inv_focal), n_shells) in cases.items():
        h = rt.hittables(items)
        scene = rt.Scene(h)
        assert scene.info["n_shells"] == n_shells, name
        # around a group of shells there is no tree, everything is tested per segment; six concentric spheres that
        # are no such group go into a tree
        assert scene.info["n_wide"] == (len(items) if n_shells else 0) and (scene.info["n_nodes"] == 0) == (n_shells > 0), name
        cam = rt.Camera.build(lookat, lookfrom, inv_focal, 0.02, 64, 48)
        spp = 4
        img, _ = scene.render(cam, spp, depth=50, seed=9)
        px = np.repeat(np.arange(64 * 48, dtype=np.uint32), spp)
        sm = np.tile(np.arange(spp, dtype=np.uint32), 64 * 48)
        cols = scene.trace_paths(cam, px, sm, depth=50, seed=9).reshape(64 * 48, spp, 3)
        total = np.zeros((64 * 48, 3), dtype=np.float32)
        for k in range(spp):
            total += cols[:, k, :]
        assert np.array_equal(total.reshape(48, 64, 3), img), name
        # the shells are in view: paths differ from the plain sky
        assert np.unique(cols.reshape(-1, 3), axis=0).shape[0] > 50, name


def test_render_multi_splits_samples_over_scene_handles(rt, scene1):
    """rt_render_multi on a one-GPU box: three handles on the same device stand in for three GPUs; the sample split,
    the concurrent renders, the peer copies and the ordered sum are the code that runs on a real node."""
    h = rt.gen_scene(1)
    scenes = [scene1, rt.Scene(h), rt.Scene(h)]
    cam = rt.default_camera(160, 96)
    s, q = rt.render_multi(scenes, cam, 10, depth=50, seed=4)
    parts = [scene1.render(cam, e - b, depth=50, seed=4, sample_begin=b)[0] for b, e in (rt.sample_shard(10, g, 3) for g in range(3))]
    assert np.array_equal(s, (parts[0] + parts[1]) + parts[2])  # handle order, f32
    whole, q_whole = scene1.render(cam, 10, depth=50, seed=4)
    assert np.allclose(s, whole, rtol=2e-6, atol=1e-6)
    assert np.abs(q.astype(int) - q_whole.astype(int)).max() <= 1 and (q != q_whole).mean() < 1e-3
    one, q_one = rt.render_multi([scene1], cam, 10, depth=50, seed=4)
    assert np.array_equal(one, whole) and np.array_equal(q_one, q_whole)
    with pytest.raises(rt.RtError):
        rt.render_multi(scenes, cam, 2)  # fewer samples than devices


def test_render_multi_second_call_allocates_nothing(rt, scene1):
    """rt_render_multi keeps its streams, frame buffers, landing slots and events for a handle list: the second call
    for the same list and frame size leaves the device's free memory as it was and takes about as long as the three
    shares rendered one after the other (three handles on the one GPU cannot overlap), not milliseconds more."""
    import ctypes as C
    import time

    import hip_mem
    from raytracer_amd import _abi

    h = rt.gen_scene(1)
    scenes = [scene1, rt.Scene(h), rt.Scene(h)]
    cam = rt.default_camera(1920, 1080)
    spp = 30
    first, _ = rt.render_multi(scenes, cam, spp, depth=50, seed=4, want_rgb8=False)
    free0 = hip_mem.mem_free()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        rt.render_multi(scenes, cam, spp, depth=50, seed=4, want_sum=False, want_rgb8=True)  # 6 MB back, not 25
        times.append(time.perf_counter() - t0)
    assert hip_mem.mem_free() == free0
    again, _ = rt.render_multi(scenes, cam, spp, depth=50, seed=4, want_rgb8=False)
    assert np.array_equal(first, again)
    # the three shares as plain device renders, one after the other
    buf = hip_mem.DeviceArray(1920 * 1080 * 3)
    stream = hip_mem.Stream()
    singles = []
    for _ in range(3):
        t0 = time.perf_counter()
        for g, sc in enumerate(scenes):
            b, e = rt.sample_shard(spp, g, 3)
            o = rt.Scene.opts(b, e, 50, 4)
            _abi.check(_abi.lib.rt_render_device(sc.handle, C.byref(cam.c), C.byref(o), buf.ptr, stream.handle))
        stream.synchronize()
        singles.append(time.perf_counter() - t0)
    # finalize + 6 MB download + two 25 MB device copies + the sum are the multi call's own work: 1 ms covers them
    assert min(times) < min(singles) + 1.5e-3, (times, singles)
    # a different frame size replaces the cache, and a destroyed scene drops it
    small, _ = rt.render_multi(scenes, rt.default_camera(64, 48), 6, depth=10, seed=1)
    assert np.isfinite(small).all()
    del scenes[2]
    s2, _ = rt.render_multi(scenes, rt.default_camera(64, 48), 6, depth=10, seed=1)
    assert np.allclose(s2, small, rtol=2e-6, atol=1e-6)


def test_render_multi_reduces_through_rccl(rt):
    """The RCCL branch of rt_render_multi with the one communicator a one-GPU box allows (RT_MULTI_REDUCE=rccl, one
    handle): librccl is found and bound, ncclCommInitAll / ncclReduce run on the render's stream, and the image that
    comes back is the plain render's.  In a process of its own: RCCL keeps threads and device memory until exit."""
    import subprocess
    import sys

    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import raytracer_amd as rt\n"
        "sc = rt.Scene(rt.gen_scene(1)); cam = rt.default_camera(160, 96)\n"
        "whole, q = sc.render(cam, 10, depth=50, seed=4)\n"
        "os.environ['RT_MULTI_REDUCE'] = 'rccl'\n"
        "try:\n"
        "    one, q1 = rt.render_multi([sc], cam, 10, depth=50, seed=4)\n"
        "except rt.RtError as e:\n"
        "    print('NO-RCCL' if 'could not be loaded' in str(e) else 'ERROR', e); sys.exit(0)\n"
        "print('EQUAL' if np.array_equal(one, whole) and np.array_equal(q1, q) else 'DIFFERENT')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a communicator that cannot be set up (the RCCL error string is in the output) or a call that does not come back
    # is a failure; only a machine without librccl skips
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    if "NO-RCCL" in out.stdout:
        pytest.skip("librccl.so cannot be loaded on this machine: " + out.stdout.strip()[-200:])
    assert "EQUAL" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_render_device_only_enqueues_once_reserved(rt, scene1):
    """After rt_scene_reserve a render call allocates nothing and waits for nothing: it returns while the kernel is
    still running (the stream is not done when the call is back), twice in a row on a stream of the caller's."""
    import ctypes as C
    import time

    import hip_mem
    from raytracer_amd import _abi

    cam = rt.default_camera(1920, 1080)
    spp = 100  # ~12 ms of kernel; the call itself takes tens of microseconds
    scene1.reserve(cam, spp)
    stream = hip_mem.Stream()
    a, b = hip_mem.DeviceArray(1920 * 1080 * 3), hip_mem.DeviceArray(1920 * 1080 * 3)
    o = rt.Scene.opts(0, spp, 50, 0)
    back = []
    for buf in (a, b):
        t0 = time.perf_counter()
        _abi.check(_abi.lib.rt_render_device(scene1.handle, C.byref(cam.c), C.byref(o), buf.ptr, stream.handle))
        back.append((time.perf_counter() - t0, stream.done()))
    stream.synchronize()
    assert stream.done()
    assert not back[0][1] and not back[1][1], back   # both calls came back before their kernels had finished
    assert max(t for t, _ in back) < 5e-3, back       # and well inside one kernel's duration
    ra, rb = a.numpy(), b.numpy()
    assert np.array_equal(ra, rb) and np.isfinite(ra).all()
    want, _ = scene1.render(cam, spp, depth=50, seed=0, want_rgb8=False)
    assert np.array_equal(ra.reshape(1080, 1920, 3), want)


def test_translated_scene_is_descended_like_the_untranslated_one(rt, ob, scene1):
    """The LDS-resident tree keeps its boxes as f16 values relative to the scene's centre and the descent works in
    those relative coordinates, so a scene far from the coordinate origin (where a half float has an ulp of 8 or more)
    is traversed with the same boxes, hence the same number of steps, as the scene at the origin."""
    from raytracer_amd import _abi

    shift = np.array([2.0e4, 0.0, 2.0e4])
    h = rt.gen_scene(1)
    moved_h = h.copy()
    moved_h["centre"] += shift
    moved, omoved = rt.Scene(moved_h), ob.Scene(moved_h)
    assert moved.info["lds_resident"] == 1 and moved.info["n_wide"] == 1
    cam = rt.default_camera(160, 96)
    cam_moved = rt.Camera.build(list(shift), list(np.array([13, 1.5, 3]) + shift), 1.0, 0.1, 160, 96)
    o, d = moved.generate_primary(cam_moved, seed=0, pixel_begin=0, pixel_end=160 * 96)
    ids, t, nrm = moved.trace_primary(o, d)
    want_ids, want_t, want_n, _ = omoved.trace(o, d, ob.TRAVERSE_GATED_BRUTE)
    _check_hits(ids, t, nrm, want_ids, want_t, want_n)

    def steps_per_path(scene, camera, spp=8):
        s, _ = scene.render(camera, spp, depth=50, seed=0, flags=_abi.RT_RENDER_STATS, want_rgb8=False)
        st = np.zeros(8, dtype=np.uint64)
        _abi.check(_abi.lib.rt_debug_render_stats(scene.handle, st.ctypes.data))
        return float(st[1]) / (160 * 96 * spp), s

    here, img = steps_per_path(scene1, cam)
    there, img_moved = steps_per_path(moved, cam_moved)
    assert abs(there - here) < 0.1 * here, (here, there)
    assert abs(img_moved.mean() - img.mean()) < 0.02 * img.mean()
    # the render kernel's paths on the moved scene against the oracle (the f32 camera ray is coarse out there, the
    # paths it starts are still the oracle's)
    got = moved.render_paths(cam_moved, 2, depth=10, seed=4).reshape(-1, 3).astype(np.float64)
    assert np.isfinite(got).all()
    px = np.repeat(np.arange(160 * 96, dtype=np.uint32), 2)
    sm = np.tile(np.arange(2, dtype=np.uint32), 160 * 96)
    assert np.array_equal(got.astype(np.float32), moved.trace_paths(cam_moved, px, sm, depth=10, seed=4))
