#!/usr/bin/env python3
"""tools/exp_hostcopy.py — device -> host rates on this box: pinned, registered, pageable; cost of hipHostRegister."""
import ctypes as C
import json
import time

import numpy as np

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
out = {}
for mib in (48, 768):
    n = mib << 20
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), n, 0) == 0

    def rate(dst, reps=5):
        hip.hipMemcpy(dst, d, n, 2)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            assert hip.hipMemcpy(dst, d, n, 2) == 0
            ts.append(time.perf_counter() - t)
        return round(n / min(ts) / 1e9, 2)
    r = {'pinned_GBps': rate(p)}
    a = np.zeros(n, np.uint8)
    r['pageable_touched_GBps'] = rate(a.ctypes.data)
    t = time.perf_counter()
    b = np.zeros(n, np.uint8)            # untouched pages (calloc)
    assert hip.hipMemcpy(b.ctypes.data, d, n, 2) == 0
    r['pageable_fresh_first_copy_ms'] = round((time.perf_counter() - t) * 1e3, 2)
    c = np.zeros(n, np.uint8)
    t = time.perf_counter()
    rc = hip.hipHostRegister(c.ctypes.data, n, 1)
    r['register_fresh_ms'] = round((time.perf_counter() - t) * 1e3, 2)
    r['register_rc'] = rc
    if rc == 0:
        r['registered_GBps'] = rate(c.ctypes.data)
        t = time.perf_counter()
        hip.hipHostUnregister(c.ctypes.data)
        r['unregister_ms'] = round((time.perf_counter() - t) * 1e3, 2)
        t = time.perf_counter()
        hip.hipHostRegister(c.ctypes.data, n, 1)
        r['register_touched_ms'] = round((time.perf_counter() - t) * 1e3, 2)
        hip.hipHostUnregister(c.ctypes.data)
    t = time.perf_counter()
    a2 = np.empty(n, np.uint8)
    np.copyto(a2, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)))
    r['host_memcpy_pinned_to_fresh_GBps'] = round(n / (time.perf_counter() - t) / 1e9, 2)
    t = time.perf_counter()
    np.copyto(a2, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)))
    r['host_memcpy_pinned_to_touched_GBps'] = round(n / (time.perf_counter() - t) / 1e9, 2)
    out['%d MiB' % mib] = r
    hip.hipFree(d)
    hip.hipHostFree(p)
print(json.dumps(out, indent=1))
