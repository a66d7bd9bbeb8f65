#!/usr/bin/env python3
"""tools/census_tiles.py [GW GH] — what a 64-pixel wavefront executes on chess @4096^2, row by row: tape ops by opcode and SKIP
tests (guards per rectangle of GW pixels x GH rows, default 64 x 32; wave-level regions) taken / not taken, with the numpy tape
evaluator of the test suite (CPU only).  DESIGN.md section 4.1 quotes it."""
import sys, collections, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import maray_amd as M, tape_eval as T
from tape_eval import OP, decode, K_YVAL, DST_NONE
data=open('/root/repo/tests/golden/chess.maray','rb').read()
s=M.Scene(data); s.rescale(4,4); tape=s.lower()
consts,row_ops,pix_ops=tape.arrays(); info=tape.info
NAMES={v:k for k,v in OP.items()}
n_ynum = 292
GW = int(sys.argv[1]) if len(sys.argv) > 2 else 64
GH = int(sys.argv[2]) if len(sys.argv) > 2 else 32
def prof_row(y, w=4096):
    # guards per rectangle of GH rows x GW px
    g0 = (y//GH)*GH
    cnt=collections.Counter()
    per_wave=[]
    for tx in range(0,w,GW):
        ys=np.array([float(y)])
        outs=T.run_section(row_ops,consts,info['n_row_slots'],None,ys,None,None,info['n_yvals'],True,w=w,span=(tx,min(w,tx+GW)-1),yspan=(np.array([float(g0)]),np.array([float(g0+GH-1)])))
        yv=np.stack(outs,axis=-1)[0]
        for x0 in range(tx,tx+GW,64):
            X=np.arange(x0,x0+64,dtype=np.float64); Y=np.full(64,float(y))
            yvb=np.broadcast_to(yv[None,:],(64,info['n_yvals']))
            c=run_count(pix_ops,consts,info['n_pix_slots'],X,Y,yvb)
            per_wave.append(sum(v for k,v in c.items() if not k.startswith('skip')))
            cnt.update(c)
    return cnt, per_wave
def run_count(ops,consts,n_slots,X,Y,yvals):
    shape=np.shape(Y); slots=[None]*max(n_slots,1); acc=None; c=collections.Counter()
    def fetch(ref):
        kind,idx=ref>>14,ref&0x3FFF
        if kind==0: return slots[idx]
        if kind==1: return np.full(shape,consts[idx])
        if kind==2: return yvals[...,idx]
        return X if idx==0 else (Y if idx==1 else acc)
    pc=-1; n=len(ops)
    with np.errstate(all='ignore'):
        while pc+1<n:
            pc+=1
            op,aux,dst,ra,rb=decode(ops[pc])
            if op==0: continue
            if op in (18,19):
                gv=fetch(ra); want=0.0 if op==18 else 1.0
                isg = (ra>>14)==2 and (ra&0x3FFF)>=n_ynum
                taken=bool(np.all(gv==want))
                c['skip_guard' if isg else 'skip_wave']+=1
                if taken:
                    c['skip_guard_taken' if isg else 'skip_wave_taken']+=1
                    acc=np.full(shape,want)
                    if dst!=DST_NONE: slots[dst]=acc
                    pc+=aux
                continue
            if op==16: c['OUT']+=1; continue
            a=fetch(ra)
            if op==1: r=a
            elif op==2: r=-a
            elif op==6: r=np.where(a>=0,1.0,0.0)
            elif op==17: r=np.where(T._sin(a)>=0,1.0,0.0)
            elif op==10: r=a+fetch(rb)
            elif op==11: r=a*fetch(rb)
            elif op==12: r=T._max(a,fetch(rb))
            elif op==13: r=T._min(a,fetch(rb))
            elif op==4: r=1.0/a
            else: raise ValueError(op)
            c[NAMES[op]]+=1
            acc=r
            if dst!=DST_NONE: slots[dst]=r
    return c
for y in (1000, 2100, 2400, 2800, 3200):
    c,pw=prof_row(y)
    tot=sum(v for k,v in c.items() if not k.startswith('skip'))
    print('row',y,'waves',len(pw),'ops/wave avg',round(tot/len(pw),1),'max',max(pw),'min',min(pw))
    print('   ',{k:round(v/len(pw),1) for k,v in sorted(c.items())})
