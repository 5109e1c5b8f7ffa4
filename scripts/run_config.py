#!/usr/bin/env python3
"""Times one of the parity/stress configurations on cuda:0 and prints one JSON line.

    python scripts/run_config.py c1|c2|c2d10|c3|c4|c5 [--spp N] [--stats]

c1 bundled scene.json 512x512 (disc + emitter + texture, linear path)   c2 random spheres 1080p depth 50
c2d10 the same at the reference's depth 10    c3 one GPU's share (125 of 1000 spp) of the same scene at 3840x2160    c4 earth + nested glass shells    c5 1M spheres 1080p 64 spp    c6 7.8M spheres (scene larger than the Infinity Cache) 1080p 16 spp
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--timing", action="store_true", help="production kernel with a cycle stamp around each phase")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--builder", default="sah", choices=["sah", "lbvh"])
    ap.add_argument("--chunk", type=int, default=0, help="samples per work item (0 = chosen by the library)")
    ap.add_argument("--kernel-ms", type=float, default=0.0, help="kernel time of an un-instrumented run, for the roofline of a --stats run")
    a = ap.parse_args()
    import numpy as np
    import torch

    import raytracer_amd as rt
    from raytracer_amd import _abi, roofline

    earth = np.load(os.path.join(ROOT, "tests", "golden", "earthmap_rgb8.npz"))["rgb8"]
    t0 = time.time()
    if a.config in ("c2", "c2d10"):
        h, tex, cam, spp, depth = rt.gen_scene(1), [], rt.default_camera(a.width, a.height), 100, (50 if a.config == "c2" else 10)
    elif a.config == "c3":  # one GPU's share of the 8-GPU configuration: samples [0, 125) of 1000 at 3840x2160
        h, tex, cam, spp, depth = rt.gen_scene(1), [], rt.default_camera(3840, 2160), 125, 50
    elif a.config == "c4":
        h, tex, spp, depth = rt.gen_scene_shells(8), [earth], 100, 50
        cam = rt.Camera.build([0, 0, 1], [0, 1.5, -5], 1.0, 0.1, 1920, 1080)
    elif a.config == "c5":
        h, tex, spp, depth = rt.gen_scene_grid(1, 1000), [], 64, 50
        cam = rt.Camera.build([0, 0, 0], [130, 40, 30], 1.0, 0.1, 1920, 1080)
    elif a.config == "c6":  # beyond the 256 MB Infinity Cache: 7.84 M spheres, 0.7 GB touched per frame, rays gather from HBM
        h, tex, spp, depth = rt.gen_scene_grid(1, 2800), [], 16, 50
        cam = rt.Camera.build([0, 0, 0], [360, 110, 80], 1.0, 0.1, 1920, 1080)
    elif a.config == "c1":
        cfg = rt.Config.from_json(os.path.join(ROOT, "tests", "golden", "scene.json"))
        h, tex, cam, spp, depth = cfg.hittables, cfg.textures, cfg.camera, 10, 10
    else:
        raise SystemExit("unknown config")
    gen_s = time.time() - t0
    spp = a.spp or spp
    scene = rt.Scene(h, tex, builder=a.builder)
    info = scene.info
    H, W = cam.vert_res, cam.horiz_res
    d_sum = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    flags = a.flags | (_abi.RT_RENDER_STATS if a.stats else 0) | (_abi.RT_RENDER_TIMING if a.timing else 0)
    scene.render_device(cam, d_sum.data_ptr(), 0, spp, depth, 0, stream=stream, flags=flags, samples_per_item=a.chunk)
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        scene.render_device(cam, d_sum.data_ptr(), 0, spp, depth, 0, stream=stream, flags=flags, samples_per_item=a.chunk)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    out = {"config": a.config, "builder": a.builder, "n_hittables": int(info["n_hittables"]), "bvh_nodes": int(info["n_nodes"]), "bvh_depth": int(info["bvh_depth"]),
           "lds_resident": int(info["lds_resident"]), "scene_device_MB": round(info["device_bytes"] / 1e6, 2),
           "scene_build_s": round(info["build_seconds"], 3), "scene_gen_s": round(gen_s, 3), "width": W, "height": H, "spp": spp, "depth": depth,
           "kernel_ms": round(min(ms), 3), "Mpaths_per_s": round(W * H * spp / min(ms) / 1e3, 1),
           "image_mean": round(float(d_sum.mean().item()) / spp, 5)}
    if a.stats:
        st = np.zeros(8, dtype=np.uint64)
        _abi.check(_abi.lib.rt_debug_render_stats(scene.handle, st.ctypes.data))
        o = st.astype(float)
        paths = W * H * spp
        seg = o[5] / paths  # one shade-phase visit per segment (+ one per fresh path that needed none)
        out["stats"] = {"inner_steps_per_path": round(o[1] / paths, 2), "leaf_tests_per_path": round(o[3] / paths, 2),
                        "shade_visits_per_path": round(seg, 3), "inner_lane_util": round(o[1] / max(o[0], 1), 3),
                        "leaf_lane_util": round(o[3] / max(o[2], 1), 3), "shade_lane_util": round(o[5] / max(o[4], 1), 3)}
        out["algorithmic_bytes_per_segment"] = roofline.bytes_per_segment(len(h))
        out["executed_node_bytes_per_path"] = round(o[1] / paths * 32 + o[3] / paths * 32, 1)  # 32-byte nodes (f16 in LDS or 16-bit grid in HBM), 32-byte spheres
        segments = (o[5] - o[6]) / paths  # every shade visit of a live path closes one segment
        out["segments_per_path"] = round(segments, 4)
        # the roofline of the resource that binds this configuration; the rate is taken from an un-instrumented run
        # of the same kernel (`--stats` slows it down), which the caller passes with --kernel-ms
        ms = a.kernel_ms or out["kernel_ms"]
        if int(info["lds_resident"]):
            fpp = roofline.flops_per_path(len(h), segments)
            ach = paths * fpp / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "valu", "achieved": round(ach, 3), "peak": roofline.PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / roofline.PEAK_FP32_TFLOPS, 4), "algorithmic_flops_per_path": round(fpp, 1), "kernel_ms": ms}
        else:
            ach = o[1] / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "l1-gather", "achieved": round(ach, 1), "peak": roofline.GATHER_GNODES_L1, "unit": "1e9 node fetches/s",
                               "frac": round(ach / roofline.GATHER_GNODES_L1, 4), "kernel_ms": ms,
                               "note": "per-lane 32-byte node gathers; peak = tools/microbench/gather_rate on an L1-resident table; "
                                       "the same walk out of L2 / Infinity Cache / HBM reaches %g / %g / %g"
                                       % (roofline.GATHER_GNODES_L2, roofline.GATHER_GNODES_MALL, roofline.GATHER_GNODES_HBM),
                               "algorithmic_bytes_per_path": round(segments * roofline.bytes_per_segment(len(h)), 1),
                               "algorithmic_TBps": round(paths * segments * roofline.bytes_per_segment(len(h)) / (ms * 1e-3) / 1e12, 2)}
    if a.timing:
        st = np.zeros(8, dtype=np.uint64)
        _abi.check(_abi.lib.rt_debug_render_stats(scene.handle, st.ctypes.data))
        o = st.astype(float)
        total = max(o[4], 1.0)
        out["timing"] = {"shade": round(o[0] / total, 4), "descent_header": round(o[1] / total, 4), "inner_loop": round(o[2] / total, 4),
                         "leaf_turn": round(o[3] / total, 4), "shade_close_segment": round(o[6] / total, 4), "shade_end_path": round(o[7] / total, 4),
                         "waves": int(o[5]), "wave_cycles_per_path": round(total / (W * H * spp), 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
