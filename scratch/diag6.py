import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, raytracer_amd as rt
h = rt.gen_scene(1); scene = rt.Scene(h); cam = rt.default_camera(160,90)
rng = np.random.default_rng(3)
pixel = np.repeat(rng.choice(160*90, 3000, replace=False), 4).astype(np.uint32); sample = np.tile(np.arange(4,dtype=np.uint32), 3000)
gi, gt, gn = scene.path_segments(cam, pixel, sample, depth=10, seed=9, max_segments=10)
k = np.nonzero(gn>=4)[0][:6]
for j in k: print(gi[j][:6], gt[j][:6], [h['albedo'][i][2] if i>=0 else None for i in gi[j][:4]])
