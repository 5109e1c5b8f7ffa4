"""`python -m raytracer_amd [-s N] [-r]`: the role of the reference's src/main.rs.

Reads ./scene.json (or builds the random scene with -r) and writes ./image.ppm.
Extra flags cover what the reference hard-codes: resolution, depth, seed, paths.
"""
import argparse
import time

from . import api


def main():
    ap = argparse.ArgumentParser(prog="raytracer")
    ap.add_argument("-s", "--samples", type=int, default=10)          # src/main.rs:65
    ap.add_argument("-r", "--random", action="store_true")            # src/main.rs:67
    ap.add_argument("--scene", default="./scene.json")                # src/main.rs:26
    ap.add_argument("--out", default="./image.ppm")                   # src/main.rs:38
    ap.add_argument("--width", type=int, default=512)                 # src/scenegen.rs:76-77
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--depth", type=int, default=api.REFERENCE_DEPTH)  # src/lib.rs:172
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    if a.random:
        cam = api.default_camera(a.width, a.height)
        scene = api.Scene(api.gen_scene(a.scene_seed), device=a.device)
        print("number of BoundingBox: %d" % scene.info["n_spheres"])  # src/scenegen.rs:65
    else:
        cfg = api.Config.from_json(a.scene)
        cam = cfg.camera
        scene = api.Scene(cfg.hittables, cfg.textures, device=a.device)
    print("Starting render...")
    print("Computing with %d samples" % (cam.horiz_res * cam.vert_res * a.samples))  # 64-bit, unlike src/main.rs:48
    t0 = time.time()
    _, rgb8 = scene.render(cam, a.samples, depth=a.depth, seed=a.seed, want_sum=False)
    api.write_ppm(a.out, rgb8)
    print("Render finished in %ds" % int(time.time() - t0))  # src/main.rs:59


if __name__ == "__main__":
    main()
