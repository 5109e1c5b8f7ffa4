#!/usr/bin/env python3
"""Warm scene-creation times of the two tree builders on the large configurations (third creation in the process, so
code objects are loaded and the allocator is warm), with the library's own stage laps (RT_CREATE_LAPS=1, on stderr).

    python scripts/scene_creation_times.py [c5|c6 ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    import raytracer_amd as rt

    sides = {"c5": 1000, "c6": 2800}
    for cfg in (sys.argv[1:] or ["c5", "c6"]):
        h = rt.gen_scene_grid(1, sides[cfg])
        for builder in ("sah", "lbvh"):
            times = []
            for rep in range(3):
                if rep == 2:
                    os.environ["RT_CREATE_LAPS"] = "1"
                    sys.stderr.write("-- %s %s\n" % (cfg, builder))
                t0 = time.perf_counter()
                scene = rt.Scene(h, builder=builder)
                times.append(time.perf_counter() - t0)
                info = scene.info
                del scene
                os.environ.pop("RT_CREATE_LAPS", None)
            print(json.dumps({"config": cfg, "builder": builder, "n_hittables": len(h), "create_ms": [round(t * 1e3, 1) for t in times],
                              "warm_create_ms": round(times[-1] * 1e3, 1), "bvh_depth": int(info["bvh_depth"])}), flush=True)


if __name__ == "__main__":
    main()
