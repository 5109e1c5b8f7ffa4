import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, raytracer_amd as rt
from raytracer_amd import _abi
scene = rt.Scene(rt.gen_scene(1)); cam = rt.default_camera(200,120)
a,_ = scene.render(cam, 6, depth=50, seed=8)
b,_ = scene.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_LOCKSTEP)
c,_ = scene.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_NO_LDS_SCENE)
for n,x in (('lds16 vs lockstep',a-b),('nolds vs lockstep',c-b)):
    bad = np.abs(x).max(axis=2)>0
    print(n, 'pixels differing', bad.sum(), 'max', np.abs(x).max(), 'means', a.mean(), b.mean(), c.mean())
    ys,xs = np.nonzero(bad); print(list(zip(ys[:10],xs[:10])))
px = np.repeat(np.arange(200*120,dtype=np.uint32),6); sm = np.tile(np.arange(6,dtype=np.uint32),200*120)
cols = scene.trace_paths(cam, px, sm, depth=50, seed=8).reshape(200*120,6,3)
acc = np.zeros((200*120,3),dtype=np.float32)
for k in range(6): acc += cols[:,k,:]
acc = acc.reshape(120,200,3)
for n,x in (('lds16',a),('lockstep',b),('nolds',c)):
    print(n,'vs paths: differing px', (np.abs(x-acc).max(axis=2)>0).sum(), np.abs(x-acc).max())
