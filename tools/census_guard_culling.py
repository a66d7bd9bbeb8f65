#!/usr/bin/env python3
"""tools/census_guard_culling.py (CPU) — chess @4096^2: which (rectangle, guard) pairs are set, and how many of a guard job's guards a wavefront of the
ROW kernel needs at all under two lane mappings (64 rectangles of one band of rows; an 8 x 8 block of rectangles).  DESIGN.md section 9."""
import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import maray_amd as M, tape_eval as T
data=open('/root/repo/tests/golden/chess.maray','rb').read()
s=M.Scene(data); s.rescale(4,4); tape=s.lower()
consts,row_ops,pix_ops=tape.arrays(); info=tape.info
W=H=4096; GW,GH=64,32
ng=H//GH; nt=W//GW
vals=np.zeros((ng,nt,info['n_yvals']))
g0=np.arange(ng,dtype=np.float64)*GH
for t in range(nt):
    outs=T.run_section(row_ops,consts,info['n_row_slots'],None,g0,None,None,info['n_yvals'],False,w=W,span=(t*GW,t*GW+GW-1),yspan=(g0,g0+GH-1))
    vals[:,t,:]=np.stack(outs,axis=-1)
n_ynum=[i for i in range(info['n_yvals'])]
# guards = y values that only gate SKIP ops: take those with values in {0,1} only and index >= numeric count (approx: last 168)
G=vals[:,:,-168:]!=0      # (ng, nt, 168)
print('set fraction', G.mean())
# mapping A: wave = one group, all 64 tiles
needA=G.any(axis=1)            # (ng,168)
# mapping B: wave = 8 groups x 8 tiles
needB=G.reshape(ng//8,8,nt//8,8,168).any(axis=(1,3))    # (16, 8, 168)
print('guards needed per wave: band mapping %.3f of all, block mapping %.3f' % (needA.mean(), needB.mean()))
# per job of 8 guards (consecutive bits approx): a job is needed if any of its guards is
jobsA=needA[:,:168//8*8].reshape(ng,-1,8).any(axis=2); jobsB=needB[:,:,:168//8*8].reshape(ng//8,nt//8,-1,8).any(axis=3)
print('jobs with any needed guard: band %.3f, block %.3f' % (jobsA.mean(), jobsB.mean()))
