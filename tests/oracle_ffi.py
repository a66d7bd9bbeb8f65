"""ctypes binding of the CPU oracle (oracle/libmaray_oracle.so).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
LIB_PATH = os.path.join(ORACLE_DIR, 'libmaray_oracle.so')


class Texture(C.Structure):
    _fields_ = [('rgb', C.c_void_p), ('w', C.c_uint32), ('h', C.c_uint32)]


_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, 'maray_oracle.c')
    if (not os.path.exists(LIB_PATH)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH)):
        build()
    L = C.CDLL(LIB_PATH)
    L.oracle_scene_from_bytes.restype = C.c_void_p
    L.oracle_scene_from_bytes.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    L.oracle_scene_free.argtypes = [C.c_void_p]
    L.oracle_last_error.restype = C.c_char_p
    L.oracle_scene_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.oracle_scene_is_legacy.argtypes = [C.c_void_p]
    L.oracle_scene_node_count.restype = C.c_uint64
    L.oracle_scene_node_count.argtypes = [C.c_void_p, C.c_int]
    L.oracle_scene_tag_histogram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.oracle_fix_color.argtypes = [C.c_void_p]
    L.oracle_scene_encode_channel.restype = C.c_size_t
    L.oracle_scene_encode_channel.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.oracle_eval2.restype = C.c_double
    L.oracle_eval2.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_uint32]
    L.oracle_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_scene_emit_c.restype = C.c_void_p
    L.oracle_scene_emit_c.argtypes = [C.c_void_p]
    L.oracle_free.argtypes = [C.c_void_p]
    L.oracle_cast_u8.restype = C.c_uint8
    L.oracle_cast_u8.argtypes = [C.c_double]
    L.oracle_op_unary.restype = C.c_double
    L.oracle_op_unary.argtypes = [C.c_int, C.c_double]
    L.oracle_op_binary.restype = C.c_double
    L.oracle_op_binary.argtypes = [C.c_int, C.c_double, C.c_double]
    _lib = L
    return L


def _tex_array(textures):
    """textures: list of HxWx3 uint8 arrays -> (ctypes array, keepalive)."""
    if not textures:
        return None, 0, []
    keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
    arr = (Texture * len(keep))()
    for i, t in enumerate(keep):
        assert t.ndim == 3 and t.shape[2] == 3
        arr[i].rgb = t.ctypes.data
        arr[i].w = t.shape[1]
        arr[i].h = t.shape[0]
    return arr, len(keep), keep


class Scene:
    def __init__(self, data, legacy=-1):
        L = lib()
        self._h = L.oracle_scene_from_bytes(data, len(data), legacy)
        if not self._h:
            raise ValueError(L.oracle_last_error().decode())

    def __del__(self):
        if getattr(self, '_h', None):
            lib().oracle_scene_free(self._h)
            self._h = None

    @property
    def size(self):
        w, h = C.c_uint32(), C.c_uint32()
        lib().oracle_scene_size(self._h, C.byref(w), C.byref(h))
        return w.value, h.value

    @property
    def legacy(self):
        return bool(lib().oracle_scene_is_legacy(self._h))

    def node_count(self, c):
        return lib().oracle_scene_node_count(self._h, c)

    def tag_histogram(self, c):
        out = (C.c_uint64 * 22)()
        lib().oracle_scene_tag_histogram(self._h, c, out)
        return list(out)

    def fix_color(self):
        lib().oracle_fix_color(self._h)

    def encode_channel(self, c):
        n = lib().oracle_scene_encode_channel(self._h, c, None, 0)
        buf = C.create_string_buffer(n)
        lib().oracle_scene_encode_channel(self._h, c, buf, n)
        return buf.raw

    def eval2(self, c, x, y, textures=None):
        arr, n, keep = _tex_array(textures)
        return lib().oracle_eval2(self._h, c, x, y, C.cast(arr, C.c_void_p) if arr else None, n)

    def render_rows(self, w, h, y0, y1, textures=None, threads=None, want_f64=True):
        arr, n, keep = _tex_array(textures)
        rows = y1 - y0
        rgb8 = np.zeros((rows, w, 3), dtype=np.uint8)
        rgb64 = np.zeros((rows, w, 3), dtype=np.float64) if want_f64 else None
        if threads is None:
            threads = os.cpu_count() or 1
        rc = lib().oracle_render_rows(self._h, w, h, y0, y1, C.cast(arr, C.c_void_p) if arr else None, n,
                                      threads, rgb8.ctypes.data, rgb64.ctypes.data if want_f64 else None)
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return rgb8, rgb64


class JitBaseline:
    """The scene compiled to native code by the system cc — stand-in for the reference's wasmer JIT
    (src/wasm.rs, src/render.rs:102-192).  CPU baseline for bench.py --cpu-jit; textures unsupported."""

    def __init__(self, scene, cache_dir=None, opt='-O2'):
        import hashlib
        import tempfile
        p = lib().oracle_scene_emit_c(scene._h)
        src = C.string_at(p)
        lib().oracle_free(p)
        d = cache_dir or os.path.join(ORACLE_DIR, '_build')
        os.makedirs(d, exist_ok=True)
        key = hashlib.sha256(src + opt.encode()).hexdigest()[:16]
        so = os.path.join(d, 'jit_%s.so' % key)
        if not os.path.exists(so):
            c = os.path.join(d, 'jit_%s.c' % key)
            with open(c, 'wb') as f:
                f.write(src)
            tmp = tempfile.mktemp(suffix='.so', dir=d)
            subprocess.check_call(['cc', opt, '-ffp-contract=off', '-fno-fast-math', '-fno-builtin-sin', '-fno-builtin-exp',
                                   '-fno-builtin-log', '-shared', '-fPIC', '-w', c, '-o', tmp, '-lm', '-lpthread'])
            os.replace(tmp, so)
        self.so = so
        self._lib = C.CDLL(so)
        self._lib.jit_render_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]

    def render_rows(self, w, y0, y1, threads=None):
        rgb8 = np.zeros((y1 - y0, w, 3), np.uint8)
        if threads is None:
            threads = os.cpu_count() or 1
        self._lib.jit_render_rows(w, y0, y1, threads, rgb8.ctypes.data)
        return rgb8


def eval1(expr, xv, yv=0.0):
    """`Expr::eval` (src/lib.rs:617-620): evaluate with Y = 0, empty Runtime."""
    from marayb import encode
    s = Scene(encode((1, 1), [expr, expr, expr]))
    return s.eval2(0, xv, yv)
