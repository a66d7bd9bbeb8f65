"""Parity fuzz: random scenes (every op, Let / Var / Arc / Decor, textures, boolean algebra), product vs oracle,
bit-exact.  CPU leg: lowering + numpy tape evaluator (also wavefront-wise with skips taken).  GPU leg: all
three back-ends through the C ABI."""
import numpy as np
import pytest

import maray_amd as M
import scenes
import tape_eval
from fuzz_scenes import scene
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64

W, H = 83, 9          # ragged: not a multiple of the wavefront


def lowered(seed, n_tex, w=W, h=H):
    data = encode((w, h), scene(seed, n_tex=n_tex))
    try:
        return data, M.Scene(data).lower()
    except M.MarayError as e:
        if e.code in (-4, -5):        # aliased ids / self reference: the reference itself is ill-defined there
            return data, None
        raise


def test_random_scenes_lowering_vs_oracle():
    tex = scenes.textures(scale=64)
    done = yspans = 0
    for seed in range(60):
        n_tex = 2 if seed % 3 == 0 else 0
        data, tape = lowered(seed, n_tex)
        if tape is None:
            continue
        t = tex if n_tex else None
        want8, want64 = OScene(data).render_rows(W, H, 0, H, t)
        got = tape_eval.render_rows(tape, W, 0, H, t)
        assert same_f64(got, want64), seed
        assert np.array_equal(tape_eval.cast_u8(got), want8), seed
        if tape.info['skip_ops']:
            assert same_f64(tape_eval.render_rows_waves(tape, W, 0, H, t), want64), seed
            assert same_f64(tape_eval.render_rows_waves(tape, W, 0, H, t, tile=64), want64), seed      # guards per span
            if tape_eval.guards_reading_y(tape)[1] == 0:                                                # ... and per group of rows
                assert same_f64(tape_eval.render_rows_waves(tape, W, 0, H, t, tile=64, yrows=4), want64), seed
                yspans += 1
        done += 1
    assert done >= 50


def _jit_contexts(cases):
    """The specialised contexts of several scenes, built side by side: a build is two compiler processes per scene
    (maray_jitc), the calling thread only waits for them.  cases: (tape, textures) pairs."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        return list(pool.map(lambda c: M.Context(c[0], textures=c[1], backend=M.BACKEND_JIT), cases))


@pytest.mark.gpu
def test_random_scenes_gpu_vs_oracle():
    tex = scenes.textures(scale=64)
    cases = []
    for seed in list(range(100, 140)) + [1007, 1145]:       # the last two: a lane mask that is EXEC itself (tools/gpu_fuzz.py found them)
        n_tex = 2 if seed % 3 == 0 else 0
        data, tape = lowered(seed, n_tex)
        if tape is not None:
            cases.append((seed, data, tape, tex if n_tex else None))
    assert len(cases) >= 30
    jit = _jit_contexts([(tape, t) for _, _, tape, t in cases])
    for (seed, data, tape, t), jctx in zip(cases, jit):
        want8, want64 = OScene(data).render_rows(W, H, 0, H, t)
        for b in (M.BACKEND_JIT, M.BACKEND_TAPE, M.BACKEND_TAPE_SMEM):
            ctx = jctx if b == M.BACKEND_JIT else M.Context(tape, textures=t, backend=b)
            got8, got64 = ctx.render_rows(W, H, 0, H)
            ctx.close()
            assert same_f64(got64, want64), (seed, b)
            assert np.array_equal(got8, want8), (seed, b)


def test_polygon_soups_lowering_vs_oracle():
    """Many textured triangles painted over one another (what examples/chess.rs builds, at random): balanced OR trees,
    group and shape guards, private regions -- evaluated with guards per row, per tile and per rectangle."""
    from fuzz_scenes import polygon_soup
    w, h = 256, 64
    for seed, kind in ((0, True), (1, True), (2, False), (3, 'colours'), (4, 'colours')):
        data = encode((w, h), polygon_soup(seed, 24, w, h, mixed=kind))
        tape = M.Scene(data).lower()
        assert tape.info['rebalanced_chains'] >= 3 and tape.info['private_regions'] >= 8
        n_guards, n_read_y = tape_eval.guards_reading_y(tape)
        assert n_guards >= (25 if kind == 'colours' else 6) and n_read_y == 0      # coloured shapes: guards on non-booleans too
        _, want64 = OScene(data).render_rows(w, h, 0, h)
        assert same_f64(tape_eval.render_rows(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=8), want64), seed


def test_curved_soups_lowering_vs_oracle():
    """The third soup family: the gating shapes are circles, boxes and rounded boxes (Sd2), cubic Bezier strokes and
    1/(1 + d^2) falloffs (fuzz_scenes.curved_soup) -- the sqrt / abs / recip / square / corner-product rules of the
    lowering's interval bounds (RowBounds::ival) decide which spans and rectangles skip a shape.  Every pixel against
    the oracle with SKIP ops ignored, taken per wavefront, per 64-pixel span, and per rectangle of 64 x 8 and 64 x 32 pixels
    (what the specialised kernels take).  profiles/r4_curved_fuzz.txt logs a sweep of 220 more."""
    from fuzz_scenes import curved_soup
    w, h = 256, 64
    for seed, kind in ((300, False), (301, True), (302, 'colours'), (303, False)):
        data = encode((w, h), curved_soup(seed, 24, w, h, mixed=kind))
        tape = M.Scene(data).lower()
        n_guards, n_read_y = tape_eval.guards_reading_y(tape)
        assert n_guards >= 8 and n_read_y == 0, (seed, n_guards, n_read_y)       # every guard holds for a rectangle
        _, want64 = OScene(data).render_rows(w, h, 0, h)
        assert same_f64(tape_eval.render_rows(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=8), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=32), want64), seed


def test_product_soups_lowering_vs_oracle():
    """Shapes whose edge tests are Steps of products, squares, roots and reciprocals of terms monotone in x and y
    (fuzz_scenes.product_soup): the lowering's rule for a product of two varying sign-definite factors, its square rule
    and the end-point bounds decide what a span or a rectangle skips.  Every pixel against the oracle with SKIP ops
    ignored, taken per wavefront, per span and per rectangle.  (A sweep of 240 more: 0 mismatches, round 4.)"""
    from fuzz_scenes import product_soup
    w, h = 192, 48
    for seed in range(500, 506):
        data = encode((w, h), product_soup(seed, 12, w, h))
        tape = M.Scene(data).lower()
        n_guards, n_read_y = tape_eval.guards_reading_y(tape)
        assert n_guards >= 10 and n_read_y == 0, (seed, n_guards, n_read_y)
        _, want64 = OScene(data).render_rows(w, h, 0, h)
        assert same_f64(tape_eval.render_rows(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=8), want64), seed
        assert same_f64(tape_eval.render_rows_waves(tape, w, 0, h, tile=64, yrows=32), want64), seed


def _soups_on_the_gpu(cases, w, h):
    """cases: (name, scene bytes).  The three evaluators agree on every pixel (u8 and f64 planes); the GUARD-FREE lowering
    of the same scene (no SKIP op, no guard, no rebalanced chain, no private region: every pixel evaluates the whole DAG
    as /root/reference/src/lib.rs:623-670 does) gives the same raster on every pixel -- the three share one lowering and
    its guards, this one shares neither; and the oracle is asked for three bands of rows drawn per scene (seeded), six
    seconds a soup for the whole raster being most of what this test used to cost."""
    import random
    tapes = [M.Scene(data).lower() for _, data in cases]
    jit = _jit_contexts([(tape, None) for tape in tapes])
    for (name, data), tape, jctx in zip(cases, tapes, jit):
        got = {}
        for b in (M.BACKEND_JIT, M.BACKEND_TAPE, M.BACKEND_TAPE_SMEM):
            ctx = jctx if b == M.BACKEND_JIT else M.Context(tape, backend=b)
            got[b] = ctx.render_rows(w, h, 0, h)
            ctx.close()
        ref8, ref64 = got[M.BACKEND_TAPE_SMEM]
        for b in (M.BACKEND_JIT, M.BACKEND_TAPE):
            assert same_f64(got[b][1], ref64), (name, b)
            assert np.array_equal(got[b][0], ref8), (name, b)
        bare = M.Scene(data).lower(skips=False)
        assert bare.info['skip_ops'] == 0 and bare.info['private_regions'] == 0
        ctx = M.Context(bare, backend=M.BACKEND_TAPE_SMEM)
        free8, free64 = ctx.render_rows(w, h, 0, h)
        ctx.close()
        assert same_f64(free64, ref64), (name, 'guard-free')
        assert np.array_equal(free8, ref8), (name, 'guard-free')
        o = OScene(data)
        rng = random.Random(name)             # (seeded by the scene's name: the same bands on every run)
        for y0 in sorted(rng.sample(range(0, h - 8), 3)):
            want8, want64 = o.render_rows(w, h, y0, y0 + 8)
            assert same_f64(ref64[y0:y0 + 8], want64), (name, y0)
            assert np.array_equal(ref8[y0:y0 + 8], want8), (name, y0)


@pytest.mark.gpu
def test_polygon_soups_gpu_vs_oracle():
    """Six soups of 70 polygons (one tree, shapes shared by two channels' trees, a colour per shape): see _soups_on_the_gpu.
    test_polygon_soups_lowering_vs_oracle holds the same family against the oracle on every pixel on the CPU."""
    from fuzz_scenes import polygon_soup
    w, h = 1024, 200
    _soups_on_the_gpu([('polygons %d' % seed, encode((w, h), polygon_soup(seed, 70, w, h, mixed=(True, False, 'colours')[seed % 3])))
                       for seed in range(10, 16)], w, h)


@pytest.mark.gpu
def test_curved_soups_gpu_vs_oracle():
    """Six soups of 60 curved shapes (circles, boxes, rounded boxes, Bezier strokes, falloffs, rings): the guards of these
    come from the sqrt / abs / recip / square interval rules; see _soups_on_the_gpu."""
    from fuzz_scenes import curved_soup
    w, h = 1024, 200
    cases = [('curved %d' % seed, encode((w, h), curved_soup(seed, 60, w, h, mixed=(True, False, 'colours')[seed % 3]))) for seed in range(400, 406)]
    for _, data in cases:
        assert tape_eval.guards_reading_y(M.Scene(data).lower())[1] == 0
    _soups_on_the_gpu(cases, w, h)


@pytest.mark.gpu
def test_product_soups_gpu_vs_oracle():
    """Six soups of 30 shapes bounded by products, squares, roots and reciprocals of monotone terms; see _soups_on_the_gpu."""
    from fuzz_scenes import product_soup
    w, h = 1024, 200
    _soups_on_the_gpu([('products %d' % seed, encode((w, h), product_soup(seed, 30, w, h))) for seed in range(600, 606)], w, h)


@pytest.mark.gpu
def test_a_thousand_triangles_gpu_vs_oracle():
    """More than 768 guarded shapes (12 guard words): a rectangle's words stay one per lane, the scene's OR tree is a
    reduction with a branch table per word, words without a bit are skipped by one ballot per pass.  The triangles carry
    patterns whose Sin arguments run wild outside the triangle: the lanes there are fed 0.0 (jit_emit.hpp, quiet_arg),
    else nearly every tile would be re-rendered by the interpreter.  Every pixel of the specialised kernels against the
    scalar-cache interpreter (u8 and f64 planes), bands against the oracle; the same scene with the tree walked as
    written (MARAY_JIT_REDUCE=0) gives the same raster."""
    import os
    from fuzz_scenes import polygon_soup
    w, h = 1536, 640
    data = encode((w, h), polygon_soup(21, 1000, w, h, mixed=False))
    tape = M.Scene(data).lower()
    assert tape.info['n_yvals'] > 3000 and tape.info['sin_bounded'] < tape.info['sin_ops']
    ref = M.Context(tape, backend=M.BACKEND_TAPE_SMEM)
    want8, want64 = ref.render_rows(w, h, 0, h)
    ref.close()
    for mode in (None, '0'):
        if mode is None:
            os.environ.pop('MARAY_JIT_REDUCE', None)
        else:
            os.environ['MARAY_JIT_REDUCE'] = mode
        try:
            ctx = M.Context(tape, backend=M.BACKEND_JIT)
            got8, got64 = ctx.render_rows(w, h, 0, h)
            ctx.close()
        finally:
            os.environ.pop('MARAY_JIT_REDUCE', None)
        assert np.array_equal(got8, want8), mode
        assert same_f64(got64, want64), mode
    o = OScene(data)
    for y0, y1 in ((0, 2), (318, 321), (638, 640)):
        o8, o64 = o.render_rows(w, h, y0, y1)
        assert np.array_equal(want8[y0:y1], o8) and same_f64(want64[y0:y1], o64), y0
