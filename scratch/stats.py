import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, raytracer_amd as rt, ctypes as C
from raytracer_amd import _abi
scene = rt.Scene(rt.gen_scene(1)); cam = rt.default_camera(1920,1080)
for fl,name in ((8,'lds16'),(12,'hbm')):
    s,_ = scene.render(cam, 20, depth=50, seed=0, flags=fl, want_rgb8=False)
    out = np.zeros(8, dtype=np.uint64); _abi.check(_abi.lib.rt_debug_render_stats(scene.handle, out.ctypes.data))
    o = out.astype(float); paths = 1920*1080*20
    print(name, 'inner turns %.3g util %.3f | leaf turns %.3g util %.3f | shade turns %.3g util %.3f regen lanes %.3g | descent turns %.3g' % (o[0]/64, o[1]/o[0], o[2]/64, o[3]/o[2], o[4]/64, o[5]/o[4], o[6], o[7]/64))
    print('   per path: inner lane-steps %.2f leaf lane-steps %.2f shades %.2f' % (o[1]/paths, o[3]/paths, o[5]/paths))
