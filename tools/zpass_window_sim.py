"""CPU simulation behind profiles/r05/experiments/zpass_windows.txt (oracle pixels; nothing of the product runs): how many points of a
Morton run fall outside an LDS window / a coarse-tile cache, and how many 64-byte z-buffer segments a block flushes.
   python tools/zpass_window_sim.py"""
import sys, numpy as np
sys.path.insert(0,'.')
from oracle import oracle
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import synth
N=1_000_000
xyz,rgb=synth.box_room(N,0)
lo,hi=xyz.min(0),xyz.max(0)
q=np.clip(((xyz-lo)/(hi-lo)*(2**21-1)).astype(np.uint64),0,2**21-1)
def spread(v):
    v=v.astype(np.uint64)
    v=(v|(v<<32))&np.uint64(0x1f00000000ffff)
    v=(v|(v<<16))&np.uint64(0x1f0000ff0000ff)
    v=(v|(v<<8))&np.uint64(0x100f00f00f00f00f)
    v=(v|(v<<4))&np.uint64(0x10c30c30c30c30c3)
    v=(v|(v<<2))&np.uint64(0x1249249249249249)
    return v
key=spread(q[:,0])|(spread(q[:,1])<<np.uint64(1))|(spread(q[:,2])<<np.uint64(2))
order=np.argsort(key,kind='stable')
P=xyz[order]
t,ypr=synth.gt_pose(0)
tr,ro=synth.start_poses(t,ypr,4,seed=0)
Hd,Wd=200,400
for PTS,TH,TW in ((2048,32,64),(4096,32,64),(2048,64,128),(1024,32,64),(2048,16,32)):
    tot_out=0; tot_cells=0; tot_seg=0; nblk=0
    for b in range(2):
        cam=synth.transform_cloud(P,tr[b],ro[b])
        row,col=oracle.pano_pixels(cam,(Hd,Wd))
        for c0i in range(0,N,PTS):
            r=row[c0i:c0i+PTS].astype(np.int64); c=col[c0i:c0i+PTS].astype(np.int64)
            m=min(PTS//2,len(r)-1)
            r0=r[m]-TH//2; c0=(c[m]-TW//2)%Wd
            tc=(c-c0)%Wd; trr=r-r0
            inside=(trr>=0)&(trr<TH)&(tc<TW)
            tot_out+=(~inside).sum()
            cells=np.unique(trr[inside]*TW+tc[inside])
            tot_cells+=len(cells)
            # 64B segments = 16 cells in a row of the z-buffer
            rr=r0+cells//TW; cc=(c0+cells%TW)%Wd
            tot_seg+=len(np.unique(rr*1000+(cc//16)))
            nblk+=1
    print("PTS %d window %dx%d: outside %.3f%% of points (%.0f per block), non-empty cells per block %.0f, 64B segments per block %.1f, per pose: out %.0f cells %.0f segs %.0f"%(PTS,TH,TW,100*tot_out/(2*N),tot_out/nblk,tot_cells/nblk,tot_seg/nblk, tot_out/2, tot_cells/2, tot_seg/2))
print("---- multi-window")
for PTS,TH,TW,NW in ((2048,32,64,2),(2048,32,64,3),(4096,32,64,2),(4096,32,64,3),(4096,32,64,4),(2048,16,32,4),(4096,64,64,2)):
    tot_out=0; tot_cells=0; tot_seg=0; nblk=0
    for b in range(2):
        cam=synth.transform_cloud(P,tr[b],ro[b])
        row,col=oracle.pano_pixels(cam,(Hd,Wd))
        for c0i in range(0,N,PTS):
            r=row[c0i:c0i+PTS].astype(np.int64); c=col[c0i:c0i+PTS].astype(np.int64)
            left=np.ones(len(r),bool)
            anchor=min(PTS//2,len(r)-1)
            for w in range(NW):
                r0=r[anchor]-TH//2; c0=(c[anchor]-TW//2)%Wd
                tc=(c-c0)%Wd; trr=r-r0
                inside=left&(trr>=0)&(trr<TH)&(tc<TW)
                cells=np.unique(trr[inside]*TW+tc[inside])
                tot_cells+=len(cells)
                rr=r0+cells//TW; cc=(c0+cells%TW)%Wd
                tot_seg+=len(np.unique(rr*1000+(cc//16)))
                left&=~inside
                if not left.any(): break
                anchor=np.nonzero(left)[0][0]
            tot_out+=left.sum(); nblk+=1
    print("PTS %d window %dx%d x%d: outside %.3f%% of points (%.1f per block), cells per block %.0f, 64B segments per block %.1f; per pose: out %.0f segs %.0f"%(PTS,TH,TW,NW,100*tot_out/(2*N),tot_out/nblk,tot_cells/nblk,tot_seg/nblk, tot_out/2, tot_seg/2))

# ---- coarse-tile cache
cams=[]
for b in range(2):
    cam=synth.transform_cloud(P,tr[b],ro[b]); cams.append(oracle.pano_pixels(cam,(Hd,Wd)))
def run(PTS,CH,CW,S,ways=1):
    tot_out=0; tot_seg=0; nblk=0; ntiles=0
    tw=(Wd+CW-1)//CW
    for b in range(2):
        row,col=cams[b]
        for c0i in range(0,N,PTS):
            r=row[c0i:c0i+PTS].astype(np.int64); c=col[c0i:c0i+PTS].astype(np.int64)
            tid=(r//CH)*tw+(c//CW)
            # direct-mapped: first come first served in point order (approximation of the race)
            uniq,first=np.unique(tid,return_index=True)
            order=np.argsort(first)
            slots={}
            ok=set()
            for u in uniq[order]:
                h=int((u*2654435761)>>7)%S if False else int(u%S)
                lst=slots.setdefault(h,[])
                if len(lst)<ways:
                    lst.append(u); ok.add(u)
            inside=np.isin(tid,list(ok))
            tot_out+=(~inside).sum()
            tot_seg+=len(np.unique(r[inside]*1000+c[inside]//16))
            ntiles+=len(uniq); nblk+=1
    print("PTS %5d coarse tile %dx%d, %d slots x %d ways (%d KB): tiles touched per block %.1f; outside %.3f%%; per pose: direct %.0f + segments %.0f = %.0f requests"%(PTS,CH,CW,S,ways,S*ways*CH*CW*4//1024,ntiles/nblk,100*tot_out/(2*N), tot_out/2, tot_seg/2,(tot_out+tot_seg)/2),flush=True)
for cfg in ((2048,8,16,16),(2048,8,16,32),(2048,8,16,64),(2048,4,16,64),(2048,8,16,16,2),(4096,8,16,64),(4096,8,16,32,2),(2048,4,16,32,2)):
    run(*cfg)
