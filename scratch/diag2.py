import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, oracle_binding as ob, raytracer_amd as rt
h = rt.gen_scene(1)
scene = rt.Scene(h); osc = ob.Scene(h)
cam = rt.default_camera(160,90); ocam = ob.camera_from_bytes(bytes(cam.c))
rng = np.random.default_rng(3)
pixel = np.repeat(rng.choice(160*90, 3000, replace=False), 4).astype(np.uint32); sample = np.tile(np.arange(4,dtype=np.uint32), 3000)
for depth in (1,2,3):
    got = scene.trace_paths(cam, pixel, sample, depth=depth, seed=9).astype(np.float64)
    for name, trav in (('gated', ob.TRAVERSE_GATED_BRUTE), ('linear', ob.TRAVERSE_LINEAR), ('tree', ob.TRAVERSE_TREE)):
        want, seg = osc.trace_paths(ocam, pixel, sample, depth=depth, seed=9, traverse=trav, with_segments=True)
        err = np.abs(got-want).max(axis=1)
        print('depth', depth, name, 'frac bad', (err>2e-4).mean(), 'max', err.max())
