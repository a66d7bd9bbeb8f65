"""Host logic of the product (reader, fix_color, Expr -> tape lowering) checked
against the oracle on the CPU.  The tape is evaluated by tests/tape_eval.py
(numpy), never by the product."""
import math

import numpy as np
import pytest

import maray_amd as M
import scenes
import tape_eval
from marayb import (add, arc, decode, decor, encode, encode_expr, let_, mul, nat, sin, step, sub, subst_xy_deep,
                    var_id, x, y, recip, div, max_, min_)
from oracle_ffi import Scene as OScene


def same_f64(a, b):
    """Bit-exact comparison of f64 planes; all NaNs compare equal (x86 and gfx950
    generate different default NaN signs; NaN never reaches a non-NaN output)."""
    a = np.asarray(a); b = np.asarray(b)
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(nan, 0, a).view(np.uint64), np.where(nan, 0, b).view(np.uint64))


def check_scene(data, w, h, rows, textures=None, hoist=True):
    tape = M.Scene(data).lower(hoist_rows=hoist)
    o = OScene(data)
    for y0, y1 in rows:
        got = tape_eval.render_rows(tape, w, y0, y1, textures)
        want8, want64 = o.render_rows(w, h, y0, y1, textures)
        assert same_f64(got, want64)
        assert np.array_equal(tape_eval.cast_u8(got), want8)
    return tape


def test_reader_matches_reference_fixture(chess_bytes):
    s = M.Scene(chess_bytes)
    assert s.size == (1024, 1024) and s.legacy
    assert [s.node_count(c) for c in range(3)] == [29314] * 3
    # save() re-encodes in the current numbering: equal to the test encoder's output
    (w, h), color = decode(chess_bytes)
    assert s.encode() == encode((w, h), color)
    # and the result is read back as a current-numbering file
    s2 = M.Scene(s.encode())
    assert not s2.legacy and s2.node_count(0) == 29314


def test_reader_errors():
    with pytest.raises(M.MarayError) as e:
        M.Scene(b'\x00' * 7)
    assert e.value.code == -3
    with pytest.raises(M.MarayError):
        M.Scene(encode((2, 2), [x(), x(), x()]) + b'\x01')
    with pytest.raises(M.MarayError) as e:
        M.Scene(path='/nonexistent/file.maray')
    assert e.value.code == -2


def test_fix_color_matches_oracle(chess_bytes):
    cases = [chess_bytes]
    nested = lambda i0, d0, i1, d1, b: let_([(i0, d0)], let_([(i1, d1)], var_id(b)))
    cases.append(encode((1, 1), [nested(0, sub(x(), nat(k)), 0, add(y(), nat(k)), 0) for k in (1, 2, 3)]))
    cases.append(encode((1, 1), [arc(nested(0, x(), 0, y(), 0)), decor(nested(0, x(), 0, y(), 0), ['t', 3]), x()]))
    for data in cases:
        s = M.Scene(data); s.fix_color()
        o = OScene(data); o.fix_color()
        (w, h), _ = decode(data)
        import struct
        assert s.encode() == struct.pack('<II', w, h) + b''.join(o.encode_channel(c) for c in range(3))


def test_chess_tape_census(chess_bytes):
    plain = M.Scene(chess_bytes).lower(skips=False).info      # no skip regions, no row bounds: the bare DAG
    # 535 constant ops folded on the host; Y-only ops hoisted
    assert plain['folded_ops'] == 535
    assert plain['alg_ops'] == plain['alg_ops_xy'] + plain['alg_ops_x'] + plain['alg_ops_y'] + plain['alg_ops_uniform']
    fused = plain['op_histogram'][tape_eval.OP['STEPSIN']]
    assert fused == 256                       # every Sin of chess feeds a Step (SURVEY.md finding 4)
    assert plain['n_pix_ops'] == plain['alg_ops_xy'] + plain['alg_ops_x'] + 3 - fused   # a fused op stands for two
    assert plain['n_row_ops'] == plain['alg_ops_y'] + plain['n_yvals']
    assert plain['n_pix_slots'] <= 32         # Sethi-Ullman order keeps few values live
    assert plain['op_histogram'][tape_eval.OP['SIN']] == 0 and plain['op_histogram'][tape_eval.OP['OUT']] == 3
    unfused = M.Scene(chess_bytes).lower(fuse=False, skips=False).info
    assert unfused['op_histogram'][tape_eval.OP['SIN']] == 256 and unfused['alg_ops'] == plain['alg_ops']
    assert plain['sin_ops'] == plain['sin_bounded'] == 256      # interval analysis: |arg| < 105414350 everywhere
    # default lowering: same computing ops + SKIP ops; row bounds add y values and ROW work
    shared = M.Scene(chess_bytes).lower(private_regions=False).info
    assert shared['folded_ops'] == 535 and shared['alg_ops'] == plain['alg_ops'] and shared['private_regions'] == 0
    assert shared['n_pix_ops'] == plain['n_pix_ops'] + shared['skip_ops']
    # ... and every row region re-derives the shared x-dependent values it reads (a few hundred more ops in the
    # tape, executed only on the rows that enter the region); the census of the scene itself is unchanged
    info = M.Scene(chess_bytes).lower().info
    assert info['folded_ops'] == 535 and info['alg_ops'] == plain['alg_ops'] and info['private_regions'] == 128
    assert shared['n_pix_ops'] < info['n_pix_ops'] < shared['n_pix_ops'] + 600
    assert info['skip_ops'] > 800 and info['n_yvals'] > plain['n_yvals'] and info['n_row_ops'] > plain['n_row_ops']
    assert info['n_pix_slots'] <= 96          # hoisting shared nodes out of skip regions costs some
    nocommute = M.Scene(chess_bytes).lower(plain_cse=True, skips=False).info
    assert nocommute['alg_ops_y'] == 844      # Y-only census of SURVEY.md §8(d)
    assert nocommute['alg_ops'] >= plain['alg_ops']


@pytest.mark.parametrize('hoist', [True, False])
def test_chess_tape_equals_oracle_on_rows(chess_bytes, hoist):
    check_scene(chess_bytes, 1024, 1024, [(0, 1), (511, 513), (600, 601), (704, 705)], hoist=hoist)


def test_skip_regions_preserve_values(chess_bytes):
    """SKIPZ / SKIPNZ taken wavefront by wavefront (as the kernels do) vs never taken vs lowering without them."""
    tape = M.Scene(chess_bytes).lower()
    assert tape.info['skip_ops'] > 500 and tape.info['bool_ops'] > 3000
    plain = M.Scene(chess_bytes).lower(skips=False)
    assert plain.info['skip_ops'] == 0
    _, want64 = OScene(chess_bytes).render_rows(1024, 1024, 600, 601)
    taken = tape_eval.render_rows_waves(tape, 1024, 600, 601)
    assert same_f64(taken, want64)
    assert same_f64(tape_eval.render_rows(tape, 1024, 600, 601), want64)
    assert same_f64(tape_eval.render_rows(plain, 1024, 600, 601), want64)
    # a scene mixing boolean algebra with arithmetic consumers of booleans, ragged width
    from marayb import max_, min_
    b1 = step(sub(x(), nat(20))); b2 = step(sub(nat(70), x())); b3 = step(sub(y(), nat(2)))
    heavy = step(sin(mul(add(mul(x(), x()), mul(y(), nat(3))), div(nat(1), nat(7)))))
    for k in range(6):
        heavy = min_(heavy, step(add(mul(x(), nat(k + 1)), sub(y(), nat(40 * k)))))
    c = [mul(mul(min_(b1, b2), heavy), nat(200)), mul(max_(mul(b1, b3), mul(heavy, sub(nat(1), b2))), add(x(), nat(1))),
         add(min_(min_(b1, b3), heavy), max_(b2, heavy))]
    data = encode((100, 5), c)
    t2 = M.Scene(data).lower()
    assert t2.info['skip_ops'] >= 2
    _, w64 = OScene(data).render_rows(100, 5, 0, 5)
    assert same_f64(tape_eval.render_rows_waves(t2, 100, 0, 5), w64)
    assert same_f64(tape_eval.render_rows_waves(t2, 100, 0, 5, tile=64), w64)     # spans cover whole wavefronts
    assert same_f64(tape_eval.render_rows(t2, 100, 0, 5), w64)


def test_guards_bounded_over_a_tile_preserve_values(chess_bytes):
    """The guards may be evaluated for any span [XMIN, XMAX] of a row (the specialised kernels use 256-pixel tiles):
    a bound that holds over the row's part skips more, and must never skip a region some pixel of the span needs."""
    s = M.Scene(chess_bytes)
    s.rescale(2, 2)                     # 2048 wide: 8 tiles of 256
    tape = s.lower()
    o = OScene(s.encode())
    for y0 in (700, 1100, 1408):
        _, want64 = o.render_rows(2048, 2048, y0, y0 + 1)
        assert same_f64(tape_eval.render_rows_waves(tape, 2048, y0, y0 + 1, tile=256), want64), y0
    _, want64 = o.render_rows(2048, 2048, 1408, 1409)
    assert same_f64(tape_eval.render_rows_waves(tape, 2048, 1408, 1409, tile=64), want64)      # any span of whole wavefronts
    # ... and for groups of rows: every guard of chess is bounded over y as well (none reads Y), so one evaluation
    # serves a rectangle of pixels
    n_guards, n_read_y = tape_eval.guards_reading_y(tape)
    assert n_guards == 168 and n_read_y == 0
    for y0 in (696, 1400):
        _, want64 = o.render_rows(2048, 2048, y0, y0 + 16)
        assert same_f64(tape_eval.render_rows_waves(tape, 2048, y0, y0 + 16, tile=256, yrows=8), want64), y0
    rowwise = s.lower(y_spans=False)
    assert tape_eval.guards_reading_y(rowwise) == (168, 168) and rowwise.info['n_row_ops'] < tape.info['n_row_ops']


def test_unfused_chess_tape_equals_oracle(chess_bytes):
    tape = M.Scene(chess_bytes).lower(fuse=False)
    _, want64 = OScene(chess_bytes).render_rows(1024, 1024, 512, 513)
    assert same_f64(tape_eval.render_rows(tape, 1024, 512, 513), want64)


def test_rescaled_chess_reproduces_original_pixels(chess_bytes):
    """Config 3 cross-check: X -> X/4, Y -> Y/4 is exact, so pixel (4i,4j) of
    the 4096^2 scene equals pixel (i,j) of the stored scene bit for bit."""
    s = M.Scene(chess_bytes)
    s.rescale(4, 4)
    assert s.size == (4096, 4096)
    big = s.lower()
    small = M.Scene(chess_bytes).lower()
    a = tape_eval.render_rows(big, 4096, 2400, 2401)[:, ::4]
    b = tape_eval.render_rows(small, 1024, 600, 601)
    assert same_f64(a, b)
    # the same rescale done by the test builders gives the same file
    (w, h), color = decode(chess_bytes)
    q = [mul(x(), recip(nat(4))), mul(y(), recip(nat(4)))]
    want = encode((4096, 4096), [subst_xy_deep(c, q) for c in color])
    assert s.encode() == want
    # and the oracle agrees with the tape on the rescaled scene
    o = OScene(want)
    _, w64 = o.render_rows(4096, 4096, 2400, 2401)
    assert same_f64(tape_eval.render_rows(big, 4096, 2400, 2401), w64)


def test_synthetic_scenes_equal_oracle():
    check_scene(encode((64, 64), scenes.radial_gradient()), 64, 64, [(0, 64)])
    check_scene(encode((96, 64), scenes.all_ops(96, 64)), 96, 64, [(0, 64)])
    tex = scenes.textures(scale=16)
    check_scene(encode((300, 40), scenes.textured(300)), 300, 40, [(0, 40)], textures=tex)


def test_corner_cases_equal_oracle():
    nanv = var_id(99)                                       # unknown variable -> NaN (src/cache.rs:40)
    c = [max_(nanv, x()), min_(mul(nanv, y()), nat(7)), step(sub(x(), y()))]
    check_scene(encode((16, 16), c), 16, 16, [(0, 16)])
    # constant-only and leaf-only channels, shared Let across channels, nested Lets
    shared = let_([(0, add(x(), nat(1))), (1, mul(var_id(0), y()))], add(var_id(1), var_id(0)))
    c = [nat(200), y(), div(shared, nat(3))]
    check_scene(encode((8, 8), c), 8, 8, [(0, 8)])
    inner = let_([(5, y())], add(var_id(5), var_id(0)))     # Var(0) unknown in the inner ctx -> NaN
    check_scene(encode((8, 8), [x(), x(), x()]), 8, 8, [(0, 8)])
    with pytest.raises(M.MarayError) as e:                  # id 0 resolves to X outside and NaN inside
        M.Scene(encode((8, 8), [let_([(0, x())], add(inner, var_id(0))), x(), x()])).lower()
    assert e.value.code == -4
    with pytest.raises(M.MarayError) as e:                  # self-referential definition
        M.Scene(encode((8, 8), [let_([(0, add(var_id(0), nat(1)))], var_id(0)), x(), x()])).lower()
    assert e.value.code == -5
    # sin of a constant is not folded on the host
    t = M.Scene(encode((8, 8), [sin(nat(1)), x(), x()])).lower()
    assert t.info['folded_ops'] == 0 and t.info['alg_ops_uniform'] == 1
    check_scene(encode((8, 8), [sin(nat(1)), x(), x()]), 8, 8, [(0, 8)])


def test_app_range_is_checked_at_context_creation():
    t = M.Scene(encode((8, 8), [scenes.textured(8)[0], x(), x()])).lower()
    assert t.info['n_app'] == 6
    with pytest.raises(M.MarayError) as e:
        M.Context(t, textures=None)
    assert e.value.code in (-6,)


def test_guards_stay_sound_where_values_overflow_or_turn_nan():
    """Shapes whose factors pass through inf and NaN inside the image (1/(x-40), exp of a large argument, 0 * inf,
    inf - inf): the static range analysis must refuse to bound those factors, and what the guards then say must
    still never hide a pixel -- per row, per tile, per rectangle."""
    w, h = 192, 24
    data = encode((w, h), scenes.shapes_through_inf_and_nan())
    tape = M.Scene(data).lower()
    n_guards, n_read_y = tape_eval.guards_reading_y(tape)
    assert n_guards >= 1
    _, want64 = OScene(data).render_rows(w, h, 0, h)
    assert same_f64(tape_eval.render_rows(tape, w, 0, h), want64)
    assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h), want64)
    assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64), want64)
    if n_read_y == 0:
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=8), want64)


def row_cone(tape, first, count):
    import ctypes as C
    L = M.lib()
    L.maray_row_cone.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    ops, n, ns = C.POINTER(C.c_uint64)(), C.c_uint32(), C.c_uint32()
    assert L.maray_row_cone(C.byref(tape.program), first, count, C.byref(ops), C.byref(n), C.byref(ns)) == 0, L.maray_last_error()
    arr = np.ctypeslib.as_array(ops, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)
    L.maray_free(ops)
    return arr, ns.value


def test_row_section_cut_by_outputs_keeps_every_value(chess_bytes):
    """Both back-ends cut the ROW section by outputs (row_split.cpp): the cone of a set of y values as a tape of its own,
    NOPs removed, SKIP regions kept, slots renumbered by liveness.  Each cone must reproduce exactly the outputs it was
    cut for -- with SKIP ops taken or not -- and need far fewer slots than the section."""
    s = M.Scene(chess_bytes)
    s.rescale(4, 4)
    tape = s.lower()
    consts, row_ops, _ = tape.arrays()
    info = tape.info
    ys = np.arange(1500, 1500 + 64, dtype=np.float64)
    span, yspan = (256, 511), (ys - ys % 8, ys - ys % 8 + 7)
    full = tape_eval.run_section(row_ops, consts, info['n_row_slots'], None, ys, None, None, info['n_yvals'], w=4096, span=span, yspan=yspan)
    worst = 0
    for first, count in ((0, 73), (73, 219), (292, 8), (300, 8), (380, 8), (452, 8), (292, 168)):
        ops, n_slots = row_cone(tape, first, count)
        if count < 100:
            worst = max(worst, n_slots)         # the cuts the interpreter makes: 8 guards to a job
        assert 0 < len(ops) < len(row_ops) and n_slots <= info['n_row_slots']
        for honor in (False, True):
            part = tape_eval.run_section(ops, consts, max(n_slots, 1), None, ys, None, None, info['n_yvals'], honor, w=4096, span=span, yspan=yspan)
            for k in range(info['n_yvals']):
                if first <= k < first + count:
                    assert part[k] is not None and same_f64(part[k], full[k]), (first, count, k, honor)
                else:
                    assert part[k] is None
    assert worst <= 72          # what an interpreter has to keep in LDS per work-item: tens of slots, not hundreds


@pytest.mark.parametrize('seed,n,mixed', [(3, 12, True), (4, 40, False), (5, 24, 'colours')])
def test_rescheduled_cones_of_random_shapes_keep_every_value(seed, n, mixed):
    """row_split.cpp's reschedule_tape on scenes other than chess: every job of 8 y values (what the interpreter's ROW and
    GUARDS kernels run) reproduces its outputs on rectangles of 8 rows x 64 pixels, SKIP ops taken or not."""
    import fuzz_scenes
    w, h = 256, 96
    tape = M.Scene(encode((w, h), fuzz_scenes.polygon_soup(seed, n, w, h, mixed=mixed))).lower()
    consts, row_ops, _ = tape.arrays()
    info = tape.info
    assert info['n_row_ops'] > 0 and info['n_yvals'] > 8
    ys = np.arange(0, h, dtype=np.float64)
    reads_y = tape_eval.guards_reading_y(tape)[1] > 0
    span, yspan = (64, 127), ((ys, ys) if reads_y else (ys - ys % 8, ys - ys % 8 + 7))
    full = tape_eval.run_section(row_ops, consts, info['n_row_slots'], None, ys, None, None, info['n_yvals'], w=w, span=span, yspan=yspan)
    for first in range(0, info['n_yvals'], 8):
        ops, n_slots = row_cone(tape, first, 8)
        for honor in (False, True):
            part = tape_eval.run_section(ops, consts, max(n_slots, 1), None, ys, None, None, info['n_yvals'], honor, w=w, span=span, yspan=yspan)
            for k in range(first, min(first + 8, info['n_yvals'])):
                if full[k] is None: continue
                assert part[k] is not None and same_f64(part[k], full[k]), (first, k, honor)


def test_sections_scheduled_side_by_side_give_the_serial_tape(chess_bytes, monkeypatch):
    """Large programs schedule their ROW and PIXEL sections on two threads (lower.cpp); the constant pool is merged so that
    the tape is, bit for bit, the one the sections give one after the other (MARAY_LOWER_SERIAL=1)."""
    from fuzz_scenes import curved_soup, polygon_soup
    scn = [chess_bytes, encode((1024, 200), polygon_soup(11, 70, 1024, 200, mixed=False)), encode((1024, 200), curved_soup(401, 60, 1024, 200, mixed='colours'))]
    for data in scn:
        s = M.Scene(data)
        monkeypatch.setenv('MARAY_LOWER_SERIAL', '1')
        a = s.lower()
        monkeypatch.setenv('MARAY_LOWER_SERIAL', '0')
        b = s.lower()
        assert a.info == b.info and a.info['n_row_ops'] + a.info['n_pix_ops'] > 2000
        for x, y in zip(a.arrays(), b.arrays()):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
