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


@pytest.mark.gpu
def test_random_scenes_gpu_vs_oracle():
    tex = scenes.textures(scale=64)
    done = 0
    for seed in list(range(100, 140)) + [1007, 1145]:       # the last two: a lane mask that is EXEC itself (tools/gpu_fuzz.py found them)
        n_tex = 2 if seed % 3 == 0 else 0
        data, tape = lowered(seed, n_tex)
        if tape is None:
            continue
        t = tex if n_tex else None
        want8, want64 = OScene(data).render_rows(W, H, 0, H, t)
        for b in (M.BACKEND_JIT, M.BACKEND_TAPE, M.BACKEND_TAPE_SMEM):
            ctx = M.Context(tape, textures=t, backend=b)
            got8, got64 = ctx.render_rows(W, H, 0, H)
            ctx.close()
            assert same_f64(got64, want64), (seed, b)
            assert np.array_equal(got8, want8), (seed, b)
        done += 1
    assert done >= 30


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


@pytest.mark.gpu
def test_polygon_soups_gpu_vs_oracle():
    from fuzz_scenes import polygon_soup
    w, h = 1024, 200
    for seed in range(10, 16):
        data = encode((w, h), polygon_soup(seed, 70, w, h, mixed=(True, False, 'colours')[seed % 3]))
        tape = M.Scene(data).lower()
        want8, want64 = OScene(data).render_rows(w, h, 0, h)
        for b in (M.BACKEND_JIT, M.BACKEND_TAPE, M.BACKEND_TAPE_SMEM):
            ctx = M.Context(tape, backend=b)
            got8, got64 = ctx.render_rows(w, h, 0, h)
            ctx.close()
            assert same_f64(got64, want64), (seed, b)
            assert np.array_equal(got8, want8), (seed, b)
