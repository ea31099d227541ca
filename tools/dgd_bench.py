import torch, time, numpy as np, sys
sys.path.insert(0, '.')
from piccolo_amd import ops, synth
N,H,W,B=1_000_000,1024,2048,32
xyz,rgb=synth.box_room(N,0); X,C=torch.from_numpy(xyz).cuda(),torch.from_numpy(rgb).cuda()
t,ypr=synth.gt_pose(0)
img=synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X,torch.from_numpy(t),torch.from_numpy(ypr)),C,(H,W)))
tr,ro=synth.start_poses(t,ypr,B,seed=0)
cloud,pano=ops.Cloud(X,C),ops.Pano(img)
box=ops.quantile_box(X,0.05)
for dm in (False,True):
    gd=ops.GradientDescent(cloud,pano,torch.from_numpy(tr).cuda(),torch.from_numpy(ro).cuda(),box,depth_mask=dm)
    gd.run(10); torch.cuda.synchronize(); t0=time.perf_counter(); gd.run(100); torch.cuda.synchronize()
    print("depth_mask",dm,"%.1f us per iteration"%((time.perf_counter()-t0)*1e4), gd.result()[:2,12].cpu().numpy())
