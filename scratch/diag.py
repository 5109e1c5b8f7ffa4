import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, oracle_binding as ob, raytracer_amd as rt
scene = rt.Scene(rt.gen_scene(1)); osc = ob.Scene(ob.gen_scene(1))
cam = rt.default_camera(160,90); ocam = ob.camera_from_bytes(bytes(cam.c))
rng = np.random.default_rng(3)
pixel = np.repeat(rng.choice(160*90, 3000, replace=False), 4).astype(np.uint32); sample = np.tile(np.arange(4,dtype=np.uint32), 3000)
got = scene.trace_paths(cam, pixel, sample, depth=10, seed=9).astype(np.float64)
want, seg = osc.trace_paths(ocam, pixel, sample, depth=10, seed=9, with_segments=True)
err = np.abs(got-want).max(axis=1)
for k in range(1,11):
    m = seg==k
    if m.any(): print(k, m.sum(), 'max err', err[m].max(), 'median', np.median(err[m]), 'frac>2e-4', (err[m]>2e-4).mean())
print('mean got/want', got.mean(), want.mean())
h = rt.gen_scene(1)
bad = (seg==2) & (err>2e-4)
good = (seg==2) & (err<=2e-4)
def first_ids(mask):
    ids=[]
    for p,s in zip(pixel[mask][:400], sample[mask][:400]):
        o,d = scene.generate_primary(cam, 9, int(p), int(p)+1, int(s), int(s)+1)
        i,_,_ = scene.trace_primary(o,d); ids.append(i[0])
    return np.array(ids)
bi = first_ids(bad); gi = first_ids(good)
print('bad first-hit materials', np.bincount(h['material'][bi], minlength=5), 'ids', np.unique(bi)[:20])
print('good first-hit materials', np.bincount(h['material'][gi], minlength=5))
k = np.nonzero(bad)[0][:8]
for j in k: print(pixel[j], sample[j], got[j], want[j], 'fuzz', h['fuzz'][first_ids(np.arange(len(seg))==j)[0]])
