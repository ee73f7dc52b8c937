"""Padding of the sparse convolutions' 16-pair MFMA chunks as a function of the output-tile height: builds the
active sets of a synthetic KITTI batch at the strides of conv2 / conv3 (submanifold 3x3x3 neighbourhoods on the cell
order the kernels use) and prints slots / pairs for tile heights 16..128 and chunk granularities 16 / 8 / 4.
CPU only.  usage: python tools/sconv_padding.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import synth
K=synth.KITTI
vs=np.array(K["voxel_size"]); r=np.array(K["point_cloud_range"])
coords=[]
for b in range(4):
    pts=synth.kitti_frame(b)[0]
    c=np.floor((pts[:,:3]-r[:3])/vs).astype(np.int64)
    ok=((c>=0)&(c<np.array([1408,1600,40]))).all(1)
    c=c[ok]
    coords.append(np.concatenate([np.full((len(c),1),b),c[:,[2,1,0]]],1))
c=np.concatenate(coords)
def down(c,s):
    # spconv stride-2 k3 pad1 produces outputs at all cells reachable; approximate by floor division then unique, plus dilation
    out=set()
    res=[]
    for dz in (0,1):
      for dy in (0,1):
        for dx in (0,1):
            q=c.copy(); q[:,1]=(q[:,1]+dz)//2; q[:,2]=(q[:,2]+dy)//2; q[:,3]=(q[:,3]+dx)//2
            res.append(q)
    q=np.unique(np.concatenate(res),axis=0)
    return q
c1=np.unique(c,axis=0); print("l1",len(c1))
c2=down(c1,2); print("l2",len(c2))
c3=down(c2,2); print("l3",len(c3))
for name,cc,shape in (("l2",c2,(21,800,704)),("l3",c3,(11,400,352))):
    Z,Y,X=shape
    lin=((cc[:,0]*(Z+2)+cc[:,1]+1)*(Y+2)+cc[:,2]+1)*(X+2)+cc[:,3]+1
    order=np.argsort(lin); lin=lin[order]
    N=len(lin)
    nb=np.zeros((27,N),bool)
    k=0
    for dz in (-1,0,1):
      for dy in (-1,0,1):
        for dx in (-1,0,1):
            t=lin+(dz*(Y+2)+dy)*(X+2)+dx
            pos=np.searchsorted(lin,t); pos[pos>=N]=N-1
            nb[k]=lin[pos]==t; k+=1
    R=nb.sum()
    print(name,"N",N,"R",R,"avg nbrs",R/N)
    for TR in (16,32,48,64,80,96,128):
        for G in (16,8,4):
            tiles=(N+TR-1)//TR
            pad=np.zeros((27,tiles*TR),bool); pad[:,:N]=nb
            cnt=pad.reshape(27,tiles,TR).sum(2)
            slots=(np.ceil(cnt/G)*G).sum()
            print("  TR %3d G %2d tiles %5d slots/R %.3f"%(TR,G,tiles,slots/R), end="")
        print()
