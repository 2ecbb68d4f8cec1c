import sys,os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from pytracer_amd import abi, flatten, scenes
from pytracer_amd.device import DeviceScene
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True)); 
for (W,H,N,D) in ((1280,720,2,5),(1280,720,3,5),(1280,720,2,8),(640,360,3,5),(1280,720,4,4)):
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=N, max_depth=D, path_state=45, path_seq=54, out_format=abi.OUT_F32)
    with DeviceScene(flat) as ds:
        ms=[]
        for _ in range(3):
            ds.render(cam, par); ms.append(ds.stats().kernel_ms)
        print(W,H,'N',N,'D',D,'kernel',ds.stats().kernel,'ms',['%.2f'%m for m in ms],'rays',ds.stats().n_rays, flush=True)
