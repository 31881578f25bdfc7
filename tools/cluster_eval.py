import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle_py as O
sc=O.build_scene('cover',1,1.5)
c=np.stack([sc.spheres['cx'],sc.spheres['cy'],sc.spheres['cz']],1).astype(np.float64); r=sc.spheres['r'].astype(np.float64)
n=len(r); med=np.median(r); big=np.where(r>4*med)[0]; small=np.where(r<=4*med)[0]
def morton_groups():
    mn=c[small].min(0); mx=c[small].max(0); ext=np.where(mx-mn>0,mx-mn,1)
    q=np.clip(((c[small]-mn)/ext*1023).astype(np.int64),0,1023); q[:, (mx-mn)==0]=0
    def spread(v):
        v=v&0x3ff; v=(v|(v<<16))&0x030000ff; v=(v|(v<<8))&0x0300f00f; v=(v|(v<<4))&0x030c30c3; v=(v|(v<<2))&0x09249249; return v
    key=spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
    order=small[np.argsort(key,kind='stable')]
    return [[b] for b in big]+[list(order[i:i+4]) for i in range(0,len(order),4)]
def kd_groups(idx=None):
    out=[]
    def rec(ids):
        if len(ids)<=4: out.append(list(ids)); return
        pts=c[ids]; ax=np.argmax(pts.max(0)-pts.min(0)); o=ids[np.argsort(pts[:,ax],kind='stable')]
        # split at a multiple of 4 nearest the middle
        h=(len(o)//2+3)//4*4
        if h>=len(o): h=len(o)//2
        rec(o[:h]); rec(o[h:])
    rec(small)
    return [[b] for b in big]+out
def stats(groups,name):
    R=[];S=[]
    for g in groups:
        pts=c[g]; C=0.5*(pts.min(0)+pts.max(0)); s=np.linalg.norm(pts-C,axis=1); R.append((s+r[g]).max()); S.append(s.max())
    R=np.array(R); print(name,'groups',len(groups),'R mean %.3f median %.3f max(small) %.3f sumR2(small) %.1f'%(R[len(big):].mean(),np.median(R[len(big):]),R[len(big):].max(),(R[len(big):]**2).sum()))
    return R
for name,fn in (('morton',morton_groups),('kd',kd_groups)):
    groups=fn(); R=stats(groups,name)
    # candidate count for sample rays: primary rays + shadow rays from floor points
    orc=O.Oracle(); orc.upload(sc)
    rng=np.random.default_rng(0); m=4000
    ijs=np.stack([rng.integers(0,1200,m),rng.integers(0,800,m),rng.integers(1,129,m)],1).astype(np.uint32)
    rays=orc.primary_rays(1200,800,ijs).astype(np.float64)
    hits=orc.closest_hit(rays.astype(np.float32))
    hitmask=hits[:,1].view(np.int32)>=0
    pos=hits[hitmask,2:5].astype(np.float64)
    L=np.ones(3)/np.sqrt(3)
    sh=np.concatenate([pos,np.tile(L,(len(pos),1))],1)
    dd=rng.normal(size=(len(pos),3)); dd/=np.linalg.norm(dd,axis=1,keepdims=True); dd[:,1]=np.abs(dd[:,1])
    sec=np.concatenate([pos,dd],1)
    for rname,rs in (('primary',rays),('shadow',sh),('diffuse',sec)):
        cnts=[]
        Cs=np.array([0.5*(c[g].min(0)+c[g].max(0)) for g in groups])
        for ray in rs[:1500]:
            o=ray[:3]; d=ray[3:]; oc=o-Cs; b=oc@d; q=(oc**2).sum(1); disc=b*b-(d@d)*(q-(R*1.03)**2)
            cnts.append((disc>0).sum())
        cnts=np.array(cnts); print('   ',rname,'cand groups/ray: mean %.2f  p90 %d  max %d   (per-half >14: %.3f)'%(cnts.mean(),np.percentile(cnts,90),cnts.max(),(cnts>20).mean()))
