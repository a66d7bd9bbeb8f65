"""tools/census_row_classes.py (CPU) — ROW ops of chess @4096 by what they depend on: the x-span (XMIN / XMAX) alone, the rows (Y, YMIN, YMAX)
alone, or both; and how many x-only / y-only values the mixed ops read.  Why the guards are not separable."""
import sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
import maray_amd as M
from tape_eval import decode, OP
s=M.Scene(open('/root/repo/tests/golden/chess.maray','rb').read()); s.rescale(4,4)
t=s.lower()
consts,row,pix=t.arrays()
n_slots=t.program.n_row_slots
slot=[None]*max(1,n_slots); acc=None
DX,DY=1,2
cls=[]; prod=[]
names={v:k for k,v in OP.items()}
def dep(ref):
    kind,idx=ref>>14,ref&0x3FFF
    if kind==0: return slot[idx]
    if kind==1: return (0,-1)
    if kind==3:
        if idx==2: return acc
        if idx in(3,4): return (DX,-1)
        if idx in(1,5,6): return (DY,-1)
    raise SystemExit('ref %d %d'%(kind,idx))
outs={}
ops=[]
for i,ins in enumerate(row):
    op,aux,dst,a,b=decode(ins)
    if op==OP['NOP']: ops.append(None); continue
    if op in(OP['SKIPZ'],OP['SKIPNZ']): ops.append(('skip',dep(a)[0])); continue
    da=dep(a)
    if op==OP['OUT']:
        outs[aux]=(da[0],da[1]); ops.append(('out',da[0])); continue
    db=dep(b) if OP['ADD']<=op<=OP['APP'] else (0,-1)
    c=da[0]|db[0]
    ops.append((names[op],c,da,db))
    acc=(c,i)
    if dst!=0xFFF: slot[dst]=(c,i)
import collections
cnt=collections.Counter(o[1] for o in ops if o and o[0] not in('out','skip'))
print('row ops by class (0 const,1 x-span,2 y,3 mixed):',cnt)
# crossing edges: mixed ops reading pure-x or pure-y producers
cross_x=set(); cross_y=set()
for i,o in enumerate(ops):
    if not o or o[0] in('out','skip'): continue
    if o[1]==3:
        for d in (o[2],o[3]):
            if d[0]==1 and d[1]>=0: cross_x.add(d[1])
            if d[0]==2 and d[1]>=0: cross_y.add(d[1])
print('distinct x-only values read by mixed ops:',len(cross_x),' y-only:',len(cross_y))
n_ynum=292
print('guards by class:',collections.Counter(v[0] for k,v in outs.items() if k>=n_ynum))
mixed_hist=collections.Counter(o[0] for o in ops if o and o[0] not in('out','skip') and o[1]==3)
print('mixed op kinds',mixed_hist)
xh=collections.Counter(o[0] for o in ops if o and o[0] not in('out','skip') and o[1]==1); print('x-only kinds',xh)
yh=collections.Counter(o[0] for o in ops if o and o[0] not in('out','skip') and o[1]==2); print('y-only kinds',yh)
