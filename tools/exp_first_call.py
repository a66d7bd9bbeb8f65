import sys, time, os
sys.path[:0]=['.','tests']
import numpy as np
import maray_amd as M
data=open('tests/golden/chess.maray','rb').read()
s=M.Scene(data); s.rescale(4,4)
# warm the process: another program's context + a render
s2=M.Scene(data); s2.rescale(2,2)
c=M.Context(s2.lower(), backend=M.BACKEND_JIT); c.render_rows(2048,2048,0,64); c.close()
pin=M.PinnedRaster(4096,4096)
tape=s.lower(); M.Context(tape, backend=M.BACKEND_JIT).close()     # code objects of the 4096 program in the process table
M.gen_cache_clear()
os.environ['MARAY_TRACE_INIT']='1'; os.environ['MARAY_TRACE_LOWER']='1'
t=time.perf_counter(); M.gen_to_image(s, backend=M.BACKEND_JIT, out=pin.array); print('first call ms', (time.perf_counter()-t)*1e3)
t=time.perf_counter(); M.gen_to_image(s, backend=M.BACKEND_JIT, out=pin.array); print('second call ms', (time.perf_counter()-t)*1e3)
