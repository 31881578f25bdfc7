import sys, numpy as np
sys.path.insert(0,'/root/repo')
from cpuraytracer_amd import HipRenderer, scenes
W,H=1200,800
sc=scenes.build_scene("cover",1,W,H); r=HipRenderer(0); r.upload(sc)
n=(W*H)>>6
pl=np.arange(n)*64+32
ijs=np.stack([pl%W, pl//W, np.ones(n)],1).astype(np.uint32)
rays=r.unit_primary_rays(W,H,ijs)
h=r.unit_closest_hit(rays)
idx=h[:,1].view(np.int32)
print("tiles",n,"miss",int((idx<0).sum()), "hit", int((idx>=0).sum()))
ty=sc.materials["type"][np.clip(idx,0,None)]
for t in range(4): print("type",t,int(((idx>=0)&(ty==t)).sum()))
