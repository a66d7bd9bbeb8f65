"""ctypes binding of libmaray_hip.so — host-side mirror of the reference's render
surface (`open`, `save`, `gen`, `gen_to_image`, `RenderMethod`; src/lib.rs:1155-1235).

Plumbing only: no arithmetic happens here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
OP_COUNT = 20
BACKEND_TAPE, BACKEND_TAPE_SMEM, BACKEND_JIT, BACKEND_AUTO = 0, 1, 2, 3
REPORT_NONE, REPORT_ROW, REPORT_DURATION_MS = 0, 1, 2


class MarayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('maray error %d: %s' % (code, msg))
        self.code = code


class Program(C.Structure):
    _fields_ = [('version', C.c_uint32), ('n_consts', C.c_uint32), ('consts', C.POINTER(C.c_double)),
                ('n_row_ops', C.c_uint32), ('row_ops', C.POINTER(C.c_uint64)), ('n_row_slots', C.c_uint32),
                ('n_yvals', C.c_uint32), ('n_pix_ops', C.c_uint32), ('pix_ops', C.POINTER(C.c_uint64)),
                ('n_pix_slots', C.c_uint32), ('n_app', C.c_uint32)]


class TapeInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ('n_consts', 'n_row_ops', 'n_row_slots', 'n_yvals', 'n_pix_ops', 'n_pix_slots', 'n_app', 'alg_ops',
                 'alg_ops_xy', 'alg_ops_x', 'alg_ops_y', 'alg_ops_uniform', 'folded_ops', 'dag_nodes',
                 'acc_operands', 'skip_ops', 'bool_ops', 'sin_ops', 'sin_bounded', 'private_regions', 'rebalanced_chains')] + [('op_histogram', C.c_uint32 * OP_COUNT)]


class Texture(C.Structure):
    _fields_ = [('rgb', C.c_void_p), ('w', C.c_uint32), ('h', C.c_uint32)]


class LowerOpts(C.Structure):
    _fields_ = [('hoist_rows', C.c_uint32), ('plain_cse', C.c_uint32), ('no_fuse', C.c_uint32), ('no_skips', C.c_uint32), ('no_row_guards', C.c_uint32),
                ('no_private_regions', C.c_uint32), ('no_rebalance', C.c_uint32), ('no_y_spans', C.c_uint32)]


class CtxOpts(C.Structure):
    _fields_ = [('backend', C.c_uint32), ('hint_mpixels', C.c_uint32), ('reserved', C.c_uint32 * 6)]


class GenOpts(C.Structure):
    _fields_ = [('backend', C.c_uint32), ('n_devices', C.c_uint32), ('tile_rows', C.c_uint32),
                ('reserved', C.c_uint32 * 5)]


class Report(C.Structure):
    _fields_ = [('kind', C.c_uint32), ('value', C.c_uint32)]


REPORT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_double)
TILE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)

_lib = None


def lib_path():
    return os.path.join(_HERE, 'libmaray_hip.so')


def lib():
    """Load libmaray_hip.so; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise MarayError(-8, 'libmaray_hip.so is not built (run `make -C maray_amd/csrc` or __graft_entry__.build())')
    L = C.CDLL(p)
    vp, u32, u64p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)
    sig = {
        'maray_last_error': (C.c_char_p, []),
        'maray_version': (C.c_char_p, []),
        'maray_scene_open': (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        'maray_scene_from_bytes': (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(vp)]),
        'maray_scene_free': (None, [vp]),
        'maray_scene_size': (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32)]),
        'maray_scene_set_size': (C.c_int, [vp, u32, u32]),
        'maray_scene_is_legacy': (C.c_int, [vp, C.POINTER(C.c_int)]),
        'maray_scene_node_count': (C.c_int, [vp, C.c_int, u64p]),
        'maray_scene_encode': (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        'maray_scene_save': (C.c_int, [vp, C.c_char_p]),
        'maray_scene_fix_color': (C.c_int, [vp]),
        'maray_scene_rescale': (C.c_int, [vp, u32, u32]),
        'maray_scene_simplify': (C.c_int, [vp]),
        'maray_scene_simplify_ex': (C.c_int, [vp, C.c_uint32]),
        'maray_scene_compress': (C.c_int, [vp, C.POINTER(u32)]),
        'maray_scene_display_len': (C.c_int, [vp, C.c_int, u64p]),
        'maray_lower': (C.c_int, [vp, C.POINTER(LowerOpts), C.POINTER(vp)]),
        'maray_tape_free': (None, [vp]),
        'maray_tape_program': (C.c_int, [vp, C.POINTER(Program)]),
        'maray_tape_get_info': (C.c_int, [vp, C.POINTER(TapeInfo)]),
        'maray_hip_device_count': (C.c_int, [C.POINTER(C.c_int)]),
        'maray_hip_ctx_create': (C.c_int, [C.c_int, C.POINTER(Program), C.POINTER(Texture), u32, C.POINTER(CtxOpts),
                                           C.POINTER(vp)]),
        'maray_hip_ctx_free': (None, [vp]),
        'maray_hip_render_rows': (C.c_int, [vp, u32, u32, u32, u32, vp, vp]),
        'maray_hip_render_tiles': (C.c_int, [vp, u32, u32, C.POINTER(u32), u32, vp, TILE_FN, vp]),
        'maray_host_alloc': (C.c_int, [C.c_size_t, C.POINTER(vp)]),
        'maray_host_free': (None, [vp]),
        'maray_host_register': (C.c_int, [vp, C.c_size_t]),
        'maray_host_unregister': (C.c_int, [vp]),
        'maray_hip_render_rows_device': (C.c_int, [vp, u32, u32, u32, u32, vp, vp, vp]),
        'maray_hip_render_blocks_device': (C.c_int, [vp, u32, u32, u32, u32, u32, u32, vp, vp, vp]),
        'maray_hip_time_rows': (C.c_int, [vp, u32, u32, u32, u32, vp, vp, C.c_int, C.POINTER(C.c_float)]),
        'maray_hip_time_blocks': (C.c_int, [vp, u32, u32, u32, u32, u32, u32, vp, vp, C.c_int, C.POINTER(C.c_float)]),
        'maray_hip_kernel_name': (C.c_char_p, [vp]),
        'maray_jit_code_key': (C.c_int, [C.POINTER(Program), C.c_char_p]),
        'maray_jit_code_cached': (C.c_int, [C.POINTER(Program), C.POINTER(C.c_int)]),
        'maray_gen_to_image': (C.c_int, [vp, C.POINTER(Texture), u32, C.POINTER(GenOpts), Report, REPORT_FN, vp, vp,
                                         u32, u32]),
        'maray_gen': (C.c_int, [vp, C.POINTER(Texture), u32, C.POINTER(GenOpts), Report, C.c_char_p]),
        'maray_gen_cache_clear': (None, []),
        'maray_gen_cache_info': (C.c_int, [C.c_char_p, C.c_size_t]),
        'maray_png_write': (C.c_int, [C.c_char_p, vp, u32, u32]),
        'maray_png_read': (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]),
        'maray_image_read': (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]),
        'maray_free': (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MarayError(rc, lib().maray_last_error().decode(errors='replace'))


def version():
    return lib().maray_version().decode()


def device_count():
    n = C.c_int(0)
    _check(lib().maray_hip_device_count(C.byref(n)))
    return n.value


def _textures(textures):
    if not textures:
        return None, 0, []
    keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
    arr = (Texture * len(keep))()
    for i, t in enumerate(keep):
        if t.ndim != 3 or t.shape[2] != 3:
            raise ValueError('textures must be HxWx3 uint8')
        arr[i].rgb = t.ctypes.data
        arr[i].w = t.shape[1]
        arr[i].h = t.shape[0]
    return arr, len(keep), keep


class Scene:
    """`([u32;2], [Expr;3])` as read by `maray::open` (src/lib.rs:1227-1235)."""

    def __init__(self, data=None, path=None):
        h = C.c_void_p()
        if path is not None:
            _check(lib().maray_scene_open(os.fsencode(path), C.byref(h)))
        else:
            _check(lib().maray_scene_from_bytes(data, len(data), C.byref(h)))
        self._h = h

    @classmethod
    def open(cls, path):
        return cls(path=path)

    def __del__(self):
        if getattr(self, '_h', None):
            lib().maray_scene_free(self._h)
            self._h = None

    @property
    def size(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(lib().maray_scene_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def set_size(self, w, h):
        _check(lib().maray_scene_set_size(self._h, w, h))

    @property
    def legacy(self):
        v = C.c_int()
        _check(lib().maray_scene_is_legacy(self._h, C.byref(v)))
        return bool(v.value)

    def node_count(self, c):
        n = C.c_uint64()
        _check(lib().maray_scene_node_count(self._h, c, C.byref(n)))
        return n.value

    def encode(self):
        n = C.c_size_t()
        _check(lib().maray_scene_encode(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib().maray_scene_encode(self._h, buf, n.value, C.byref(n)))
        return buf.raw

    def save(self, path):
        _check(lib().maray_scene_save(self._h, os.fsencode(path)))

    def fix_color(self):
        _check(lib().maray_scene_fix_color(self._h))

    def simplify(self, merge_divisors=False):
        """Expr::simplify on each channel (authoring-time rewrite rules of the reference).  merge_divisors: the one rewrite
        the reference lacks, without which its rules do not terminate on examples/chess.rs (maray_hip.h)."""
        if merge_divisors:
            _check(lib().maray_scene_simplify_ex(self._h, 1))
        else:
            _check(lib().maray_scene_simplify(self._h))

    def compress(self):
        """Expr::compress on each channel (authoring-time: names repeated sub-expressions); returns the variables introduced."""
        n = (C.c_uint32 * 3)()
        _check(lib().maray_scene_compress(self._h, n))
        return list(n)

    def display_len(self, c):
        """Characters of the reference's printed form of channel c."""
        n = C.c_uint64()
        _check(lib().maray_scene_display_len(self._h, c, C.byref(n)))
        return n.value

    def rescale(self, sx, sy):
        _check(lib().maray_scene_rescale(self._h, sx, sy))

    def lower(self, hoist_rows=True, plain_cse=False, fuse=True, skips=True, row_guards=True, private_regions=True, rebalance=True,
              y_spans=True):
        return Tape(self, hoist_rows, plain_cse, fuse, skips, row_guards, private_regions, rebalance, y_spans)


class Tape:
    """Lowered program (include/maray_tape.h)."""

    def __init__(self, scene, hoist_rows=True, plain_cse=False, fuse=True, skips=True, row_guards=True, private_regions=True, rebalance=True,
                 y_spans=True):
        o = LowerOpts()
        o.hoist_rows = 1 if hoist_rows else 0
        o.plain_cse = 1 if plain_cse else 0
        o.no_fuse = 0 if fuse else 1
        o.no_skips = 0 if skips else 1
        o.no_row_guards = 0 if row_guards else 1
        o.no_private_regions = 0 if private_regions else 1
        o.no_rebalance = 0 if rebalance else 1
        o.no_y_spans = 0 if y_spans else 1
        h = C.c_void_p()
        _check(lib().maray_lower(scene._h, C.byref(o), C.byref(h)))
        self._h = h
        self.program = Program()
        _check(lib().maray_tape_program(self._h, C.byref(self.program)))
        ti = TapeInfo()
        _check(lib().maray_tape_get_info(self._h, C.byref(ti)))
        self.info = {n: getattr(ti, n) for n, _ in TapeInfo._fields_ if n != 'op_histogram'}
        self.info['op_histogram'] = list(ti.op_histogram)

    def __del__(self):
        if getattr(self, '_h', None):
            lib().maray_tape_free(self._h)
            self._h = None

    @property
    def jit_code_key(self):
        """Cache key of this program's specialised kernels (generated sources + options + hiprtc version)."""
        buf = C.create_string_buffer(33)
        _check(lib().maray_jit_code_key(C.byref(self.program), buf))
        return buf.value.decode()

    @property
    def jit_code_cached(self):
        v = C.c_int()
        _check(lib().maray_jit_code_cached(C.byref(self.program), C.byref(v)))
        return bool(v.value)

    def arrays(self):
        """(consts, row_ops, pix_ops) as numpy copies."""
        p = self.program
        consts = np.ctypeslib.as_array(p.consts, shape=(p.n_consts,)).copy()
        row = np.ctypeslib.as_array(p.row_ops, shape=(p.n_row_ops,)).copy() if p.n_row_ops else np.zeros(0, np.uint64)
        pix = np.ctypeslib.as_array(p.pix_ops, shape=(p.n_pix_ops,)).copy() if p.n_pix_ops else np.zeros(0, np.uint64)
        return consts, row, pix


class PinnedRaster:
    """An (h, w, 3) uint8 raster in pinned host memory (maray_host_alloc): what a caller that wants the PCIe rate
    hands to the render entry points; `.array` is a numpy view of it, valid until close()."""

    def __init__(self, h, w):
        p = C.c_void_p()
        _check(lib().maray_host_alloc(h * w * 3, C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h, w, 3))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if getattr(self, '_p', None):
            self.array = None
            lib().maray_host_free(self._p)
            self._p = None

    def __del__(self):
        self.close()


class Context:
    """Device context: tape + constants + textures resident in HBM."""

    def __init__(self, tape, textures=None, device=0, backend=BACKEND_TAPE, hint_mpixels=0):
        arr, n, keep = _textures(textures)
        o = CtxOpts()
        o.backend = backend
        o.hint_mpixels = hint_mpixels
        h = C.c_void_p()
        _check(lib().maray_hip_ctx_create(device, C.byref(tape.program), arr, n, C.byref(o), C.byref(h)))
        self._h = h
        self._tape = tape

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, '_h', None):
            lib().maray_hip_ctx_free(self._h)
            self._h = None

    def render_rows(self, w, h, y0, y1, want_u8=True, want_f64=True):
        rows = y1 - y0
        rgb8 = np.zeros((rows, w, 3), np.uint8) if want_u8 else None
        rgb64 = np.zeros((rows, w, 3), np.float64) if want_f64 else None
        _check(lib().maray_hip_render_rows(self._h, w, h, y0, y1, rgb8.ctypes.data if want_u8 else None,
                                           rgb64.ctypes.data if want_f64 else None))
        return rgb8, rgb64

    def render_rows_into(self, w, h, y0, y1, rgb8):
        """Rows [y0, y1) into the caller's (y1-y0, w, 3) uint8 array (pinned or pageable)."""
        assert rgb8.dtype == np.uint8 and rgb8.flags['C_CONTIGUOUS'] and rgb8.size == (y1 - y0) * w * 3
        _check(lib().maray_hip_render_rows(self._h, w, h, y0, y1, rgb8.ctypes.data, None))

    def render_tiles(self, w, h, tiles, image, on_tile=None):
        """Row ranges [(y0, y1), ...] into one (h, w, 3) uint8 raster of the whole image, pipelined (maray_hip_render_tiles)."""
        assert image.dtype == np.uint8 and image.flags['C_CONTIGUOUS'] and image.size == h * w * 3
        flat = (C.c_uint32 * (2 * len(tiles)))(*[v for t in tiles for v in t])
        cb = TILE_FN((lambda user, a, b: on_tile(a, b)) if on_tile else 0)
        _check(lib().maray_hip_render_tiles(self._h, w, h, flat, len(tiles), image.ctypes.data, cb, None))

    def render_rows_device(self, w, h, y0, y1, d_rgb8=0, d_rgb64=0, stream=0):
        _check(lib().maray_hip_render_rows_device(self._h, w, h, y0, y1, d_rgb8 or None, d_rgb64 or None, stream or None))

    def render_blocks_device(self, w, h, y0, block_rows, block_stride, n_blocks, d_rgb8=0, d_rgb64=0, stream=0):
        """n_blocks blocks of block_rows rows, block_stride apart, first at y0, in ONE launch; outputs packed."""
        _check(lib().maray_hip_render_blocks_device(self._h, w, h, y0, block_rows, block_stride, n_blocks,
                                                    d_rgb8 or None, d_rgb64 or None, stream or None))

    def time_rows(self, w, h, y0, y1, d_rgb8=0, d_rgb64=0, reps=5):
        ms = C.c_float()
        _check(lib().maray_hip_time_rows(self._h, w, h, y0, y1, d_rgb8 or None, d_rgb64 or None, reps, C.byref(ms)))
        return ms.value

    def time_blocks(self, w, h, y0, block_rows, block_stride, n_blocks, d_rgb8=0, d_rgb64=0, reps=5):
        """time_rows for the launch render_blocks_device issues."""
        ms = C.c_float()
        _check(lib().maray_hip_time_blocks(self._h, w, h, y0, block_rows, block_stride, n_blocks, d_rgb8 or None, d_rgb64 or None,
                                           reps, C.byref(ms)))
        return ms.value

    @property
    def kernel_name(self):
        return lib().maray_hip_kernel_name(self._h).decode()


def gen_to_image(scene, size=None, textures=None, backend=BACKEND_AUTO, n_devices=0, tile_rows=0, report=None,
                 report_kind=REPORT_NONE, report_value=0, out=None):
    """`gen_to_image` (src/lib.rs:1177-1195) with RenderMethod::Hip → HxWx3 uint8 (into `out` when given)."""
    w, h = size if size else scene.size
    img = np.zeros((h, w, 3), np.uint8) if out is None else out
    assert img.shape == (h, w, 3) and img.dtype == np.uint8 and img.flags['C_CONTIGUOUS']
    arr, n, keep = _textures(textures)
    go = GenOpts()
    go.backend, go.n_devices, go.tile_rows = backend, n_devices, tile_rows

    def _cb(user, p, cw, ch, progress):
        if report:
            report(img, progress)
    cb = REPORT_FN(_cb)
    _check(lib().maray_gen_to_image(scene._h, arr, n, C.byref(go), Report(report_kind, report_value), cb, None,
                                    img.ctypes.data, w, h))
    return img


def gen(scene, path, textures=None, backend=BACKEND_AUTO, n_devices=0, report_kind=REPORT_NONE, report_value=0):
    """`gen` (src/lib.rs:1199-1213): render and write a PNG."""
    arr, n, keep = _textures(textures)
    go = GenOpts()
    go.backend, go.n_devices = backend, n_devices
    _check(lib().maray_gen(scene._h, arr, n, C.byref(go), Report(report_kind, report_value), os.fsencode(path)))


def gen_cache_clear():
    """Frees the tapes and contexts maray_gen_to_image keeps between calls."""
    lib().maray_gen_cache_clear()


def gen_cache_info():
    """One line per idle context maray_gen_to_image keeps: program key, device, kernel, the size it was chosen for."""
    buf = C.create_string_buffer(1 << 16)
    _check(lib().maray_gen_cache_info(buf, len(buf)))
    return buf.value.decode().splitlines()


def png_write(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    _check(lib().maray_png_write(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]))


def image_read(path):
    """`image::open(path).to_rgb8()`: a texture file in any format the library reads -> HxWx3 uint8."""
    return png_read(path, _fn='maray_image_read')


def png_read(path, _fn='maray_png_read'):
    p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    _check(getattr(lib(), _fn)(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    try:
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib().maray_free(p)
    return a
