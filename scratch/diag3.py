import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, oracle_binding as ob, raytracer_amd as rt
h = rt.gen_scene(1)
scene = rt.Scene(h); osc = ob.Scene(h)
cam = rt.default_camera(160,90); ocam = ob.camera_from_bytes(bytes(cam.c))
rng = np.random.default_rng(3)
pixel = np.repeat(rng.choice(160*90, 3000, replace=False), 4).astype(np.uint32); sample = np.tile(np.arange(4,dtype=np.uint32), 3000)
gi, gt, gn = scene.path_segments(cam, pixel, sample, depth=10, seed=9, max_segments=10)
wi, wt, wn = osc.path_segments(ocam, pixel, sample, depth=10, seed=9, max_segments=10)
same = np.all(gi==wi, axis=1)
print('identical id sequences:', same.mean())
bad = np.nonzero(~same)[0]
first_div = np.array([np.nonzero(gi[j]!=wi[j])[0][0] for j in bad])
print('first divergence segment histogram', np.bincount(first_div))
for j in bad[:12]:
    k = first_div[j==bad][0]
    print(pixel[j], sample[j], 'dev', gi[j][:5], gt[j][:4], '| orc', wi[j][:5], wt[j][:4], 'mat prev', h['material'][wi[j][k-1]] if k>0 else None)
