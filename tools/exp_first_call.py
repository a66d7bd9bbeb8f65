"""Where the first maray_gen_to_image call of a scene spends its time (MARAY_TRACE_INIT / MARAY_TRACE_LOWER on stderr), in a
process that is warm (HIP initialised, the program's code objects in the process table): what an animation's first frame
of a new scene costs.  Then the later calls: pinned raster, pageable raster (registered per call), registration kept."""
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
for trial in range(3):
    M.gen_cache_clear()
    if trial == 2: os.environ['MARAY_TRACE_INIT']='1'; os.environ['MARAY_TRACE_LOWER']='1'
    t=time.perf_counter(); M.gen_to_image(s, backend=M.BACKEND_JIT, out=pin.array); print('first call ms', (time.perf_counter()-t)*1e3)
    os.environ.pop('MARAY_TRACE_INIT', None); os.environ.pop('MARAY_TRACE_LOWER', None)
    t=time.perf_counter(); M.gen_to_image(s, backend=M.BACKEND_JIT, out=pin.array); print('second call ms', (time.perf_counter()-t)*1e3)
t=time.perf_counter(); M.Scene(data).lower(); print('lowering of the stored scene alone ms', (time.perf_counter()-t)*1e3)
t=time.perf_counter(); s.lower(); print('lowering of the 4096 scene alone ms', (time.perf_counter()-t)*1e3)
page=np.zeros((4096,4096,3),np.uint8)
for kw in ({},):
    for i in range(4):
        t=time.perf_counter(); M.gen_to_image(s, backend=M.BACKEND_JIT, out=page, **kw); print('pageable', kw, 'call', i, 'ms', (time.perf_counter()-t)*1e3)
    assert np.array_equal(page, pin.array)
M.gen_cache_clear()
tx=[np.random.default_rng(1).integers(0,256,(1024,1024,3),dtype=np.uint8), np.random.default_rng(2).integers(0,256,(512,2048,3),dtype=np.uint8)]
import scenes
from marayb import encode
st=M.Scene(encode((4096,4096),scenes.textured(4096)))
for i in range(4):
    t=time.perf_counter(); M.gen_to_image(st, textures=tx, backend=M.BACKEND_JIT, out=pin.array); print('textured (6 MiB of textures hashed per call) call', i, 'ms', (time.perf_counter()-t)*1e3)
M.gen_cache_clear()
