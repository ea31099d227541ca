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
# depth-masked GD iteration at cfg 2 (1M points, 2048x1024, 32 candidates): plain, mask at every iteration, and with the refresh
# bound (cfg depth_refresh_t / depth_refresh_r: ~0.3 px of a 2048-wide panorama for a point 1 m away) — the reference's schedule:
# lr 0.1, patience 5, factor 0.8, 100 iterations from the bench's starting poses
for name,kw in (("plain",dict(depth_mask=False)),("mask every iteration",dict(depth_mask=True)),
                ("mask, refresh bound 3e-3 m / 3e-3 rad",dict(depth_mask=True,depth_refresh_t=3e-3,depth_refresh_r=3e-3)),
                ("mask, refresh bound 2e-2 m / 2e-2 rad",dict(depth_mask=True,depth_refresh_t=2e-2,depth_refresh_r=2e-2)),
                ("mask every 4th iteration",dict(depth_mask=True,depth_every=4)),
                ("mask every 4th iteration + bound 3e-3",dict(depth_mask=True,depth_every=4,depth_refresh_t=3e-3,depth_refresh_r=3e-3))):
    ts=[]
    for rep in range(3):
        gd=ops.GradientDescent(cloud,pano,torch.from_numpy(tr).cuda(),torch.from_numpy(ro).cuda(),box,lr=0.1,patience=5,factor=0.8,**kw)
        torch.cuda.synchronize(); t0=time.perf_counter(); gd.run(100); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e4)
    res=gd.result().cpu().numpy(); k=int(np.argmin(res[:,12]))
    te,re=synth.pose_errors(res[k,:3],ops.rot_from_ypr(torch.from_numpy(res[k:k+1,3:6]))[0].cpu().numpy(),t,synth.rot_from_ypr_np(ypr))
    print("%-42s %.1f us per iteration (whole 100-iteration refinement / 100) | masks computed per candidate: mean %.1f | t_err %.4f m r_err %.3f deg"%(
        name,float(np.median(ts)),float(gd.depth_refresh_counts().float().mean()),te,re))
