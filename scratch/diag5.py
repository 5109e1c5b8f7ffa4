import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, oracle_binding as ob, raytracer_amd as rt
h = rt.gen_scene(1)
scene = rt.Scene(h); osc = ob.Scene(h)
cam = rt.default_camera(160,90); ocam = ob.camera_from_bytes(bytes(cam.c))
rng = np.random.default_rng(3)
pixel = np.repeat(rng.choice(160*90, 3000, replace=False), 4).astype(np.uint32); sample = np.tile(np.arange(4,dtype=np.uint32), 3000)
got = scene.trace_paths(cam, pixel, sample, depth=10, seed=9).astype(np.float64)
want, seg = osc.trace_paths(ocam, pixel, sample, depth=10, seed=9, with_segments=True)
err = np.abs(got-want).max(axis=1)
bad = np.nonzero(err>2e-4)[0]
print(len(bad), 'bad; by segments', np.bincount(seg[bad]))
for j in bad[:8]: print(seg[j], got[j], want[j])
gi, gt, gn = scene.path_segments(cam, pixel, sample, depth=10, seed=9, max_segments=10)
wi, wt, wn = osc.path_segments(ocam, pixel, sample, depth=10, seed=9, max_segments=10)
for j in bad[:5]: print('dev', gi[j][:6], gn[j], 'orc', wi[j][:6], wn[j], 'mat', h['material'][wi[j][0]], h['albedo'][wi[j][0]], h['albedo'][wi[j][1]] if wi[j][1]>=0 else None)
s1,_ = scene.render(cam, 4, depth=10, seed=9)
px = np.repeat(np.arange(160*90,dtype=np.uint32),4); sm = np.tile(np.arange(4,dtype=np.uint32),160*90)
cols = scene.trace_paths(cam, px, sm, depth=10, seed=9).reshape(-1,4,3).astype(np.float32)
acc = np.zeros((160*90,3),np.float32)
for k in range(4): acc += cols[:,k,:]
print('render vs trace_paths sum max diff', np.abs(acc.reshape(90,160,3)-s1).max())
w,_,_ = osc.render(ocam, spp=4, depth=10, seed=9)
print('render mean', s1.mean(), 'oracle mean', w.mean(), 'paths mean', acc.mean())
