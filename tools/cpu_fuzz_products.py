import sys, time
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
import maray_amd as M, tape_eval
from fuzz_scenes import product_soup
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64
w,h=192,48
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    data=encode((w,h),product_soup(seed,12,w,h))
    tape=M.Scene(data).lower()
    ng,nry=tape_eval.guards_reading_y(tape)
    _,want=OScene(data).render_rows(w,h,0,h)
    r=[same_f64(tape_eval.render_rows(tape,w,0,h),want), same_f64(tape_eval.render_rows_waves(tape,w,0,h),want), same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64),want)]
    if nry==0: r+= [same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64,yrows=8),want), same_f64(tape_eval.render_rows_waves(tape,w,0,h,tile=64,yrows=32),want)]
    bad+= not all(r)
    print(seed, tape.info['n_pix_ops'], 'guards',ng,'read y',nry, r, 'nonzero %.2f'%float((want[...,0]>0).mean()), flush=True)
print('mismatching', bad)
