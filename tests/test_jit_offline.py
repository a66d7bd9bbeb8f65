"""The hiprtc back-end generates and compiles gfx950 code without a GPU."""
import ctypes as C

import maray_amd as M
import scenes
from marayb import encode


def build(tape):
    L = M.lib()
    L.maray_jit_source.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.maray_jit_build.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    src = C.c_void_p()
    assert L.maray_jit_source(C.byref(tape.program), C.byref(src)) == 0, L.maray_last_error()
    text = C.string_at(src).decode()
    L.maray_free(src)
    code, n = C.c_void_p(), C.c_size_t()
    assert L.maray_jit_build(C.byref(tape.program), C.byref(code), C.byref(n)) == 0, L.maray_last_error().decode()
    blob = C.string_at(code, n.value)
    L.maray_free(code)
    return text, blob


def test_all_ops_scene_compiles_for_gfx950():
    tape = M.Scene(encode((64, 64), scenes.all_ops(64, 64))).lower()
    text, blob = build(tape)
    assert 'maray_jit_pixels' in text and 'mr_sin' in text and 'mr_exp(' in text and 'mr_ln(' in text
    assert blob[:4] == b'\x7fELF' and len(blob) > 4096


def test_chess_uses_the_pure_bounded_step_sin(chess_bytes):
    tape = M.Scene(chess_bytes).lower()
    assert tape.info['sin_ops'] == 256 and tape.info['sin_bounded'] == 256
    L = M.lib()
    L.maray_jit_source.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    src = C.c_void_p()
    assert L.maray_jit_source(C.byref(tape.program), C.byref(src)) == 0
    text = C.string_at(src).decode()
    L.maray_free(src)
    assert text.count('mr_stepsin_bounded_b(') == 256 and 'mr_stepsin_fast(' not in text
    assert text.count('const mr_mask ') > 2500       # half of chess is boolean algebra on lane masks (SGPR pairs)
    assert text.count('mr_mask bv') > 3600 and ' bool ' not in text.split('maray_jit_pixels')[1]


def test_row_section_is_cut_into_chunks(chess_bytes):
    """The ROW kernel evaluates independent chunks of the ROW section side by side (blockIdx.y); every y value is
    written by exactly one chunk, and the source builds."""
    import re
    s = M.Scene(chess_bytes)
    s.rescale(4, 4)
    tape = s.lower()
    L = M.lib()
    L.maray_jit_source_rows.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    src, k = C.c_void_p(), C.c_uint32()
    assert L.maray_jit_source_rows(C.byref(tape.program), C.byref(src), C.byref(k)) == 0, L.maray_last_error()
    text = C.string_at(src).decode()
    L.maray_free(src)
    assert 2 <= k.value <= 16 and text.count('    case ') == k.value
    written = sorted(int(m) for m in re.findall(r'yout\[(\d+)\] = ', text))
    assert written == list(range(tape.info['n_yvals']))
    # chunks share little: the ops emitted over all chunks stay close to the section's own count
    emitted = len(re.findall(r'const (?:double|mr_mask) ', text))
    assert emitted < 1.5 * tape.info['n_row_ops']
    build(tape)        # compiles the ROW kernel as well
