import sys, time
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
import maray_amd as M, tape_eval
from fuzz_scenes import curved_soup
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64
w,h=256,64
lo,hi=int(sys.argv[1]),int(sys.argv[2])
bad=0
for seed in range(lo,hi):
    kind=(False,True,'colours')[seed%3]
    t0=time.time()
    data=encode((w,h),curved_soup(seed,24,w,h,mixed=kind))
    tape=M.Scene(data).lower()
    info=tape.info
    ng,nry=tape_eval.guards_reading_y(tape)
    _,want=OScene(data).render_rows(w,h,0,h)
    a=same_f64(tape_eval.render_rows(tape,w,0,h),want)
    b=same_f64(tape_eval.render_rows_waves(tape,w,0,h),want)
    b2=same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64),want)
    c=same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64,yrows=8),want) if nry==0 else None
    c2=same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64,yrows=32),want) if nry==0 else None
    ok = a and b and b2 and c is not False and c2 is not False
    bad += not ok
    print(seed,kind,{k:info[k] for k in('n_row_ops','n_yvals','n_pix_ops','skip_ops','private_regions')},'guards',ng,'read y',nry,a,b,b2,c,c2,'%.1fs'%(time.time()-t0), flush=True)
print('mismatching soups:', bad)
