import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, raytracer_amd as rt, oracle_binding as ob
from raytracer_amd import _abi
h = rt.gen_scene(1)
scene = rt.Scene(h); cam = rt.default_camera(200,120); osc = ob.Scene(h); ocam = ob.camera_from_bytes(bytes(cam.c))
w,_,_ = osc.render(ocam, spp=6, depth=50, seed=8)
a,_ = scene.render(cam, 6, depth=50, seed=8, samples_per_item=6)
b,_ = scene.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_LOCKSTEP, samples_per_item=6)
c,_ = scene.render(cam, 6, depth=50, seed=8, flags=_abi.RT_RENDER_KERNEL_NO_LDS_SCENE, samples_per_item=6)
px = np.repeat(np.arange(200*120,dtype=np.uint32),6); sm = np.tile(np.arange(6,dtype=np.uint32),200*120)
cols = scene.trace_paths(cam, px, sm, depth=50, seed=8).reshape(200*120,6,3)
acc = np.zeros((200*120,3),dtype=np.float32)
for k in range(6): acc += cols[:,k,:]
acc = acc.reshape(120,200,3)
for n,x in (('ws-lds16',a),('lockstep',b),('ws-hbm',c),('trace_paths',acc)):
    d = np.abs(x-w).max(axis=2)
    print('%-12s mean %.5f  px off oracle by >1e-3: %d  mean|diff| %.5f' % (n, x.mean(), (d>1e-3).sum(), d.mean()))
print('oracle mean', w.mean())
for n,x,y in (('ws vs lockstep',a,b),('ws vs hbm',a,c),('lockstep vs paths',b,acc),('ws vs paths',a,acc)):
    print(n, (np.abs(x-y).max(axis=2)>0).sum())
