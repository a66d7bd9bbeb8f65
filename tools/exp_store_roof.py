#!/usr/bin/env python3
"""tools/exp_store_roof.py — what this box writes at best: hipMemsetAsync of 48 MiB (one RGB8 4096^2 raster) and of 384 MiB,
HIP-event timed.  The sky crop of the PIXEL kernel is set beside it in DESIGN.md section 7."""
import ctypes as C
import json

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
out = {}
for mib in (48, 384):
    n = mib << 20
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0
    e0, e1 = C.c_void_p(), C.c_void_p()
    hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
    for _ in range(5):
        hip.hipMemsetAsync(d, 7, n, None)
    hip.hipEventRecord(e0, None)
    for _ in range(50):
        hip.hipMemsetAsync(d, 7, n, None)
    hip.hipEventRecord(e1, None)
    hip.hipEventSynchronize(e1)
    ms = C.c_float()
    hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    out['%d MiB' % mib] = {'us_per_fill': round(ms.value / 50 * 1e3, 2), 'TB_s': round(n / (ms.value / 50 * 1e-3) / 1e12, 2)}
print(json.dumps(out))
