"""C++ scene builders (include/maray_builders.hpp) against the test builders and, for the
examples/chess.rs reconstruction, against the reference's published picture."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import maray_amd as M
import scenes
import tape_eval
from conftest import GOLDEN
from marayb import div, encode, mul, nat, x


@pytest.fixture(scope='module')
def scene_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp('scenes'))
    exe = os.path.join(os.path.dirname(M.lib_path()), 'maray_scenes')
    subprocess.check_call([exe, d])
    return d


def test_cpp_builders_write_the_same_files_as_the_test_builders(scene_dir):
    want = {'radial_1024.maray': encode((1024, 1024), scenes.radial_gradient()),
            'allops_4096.maray': encode((4096, 4096), scenes.all_ops(4096, 4096)),
            'textured_4096.maray': encode((4096, 4096), scenes.textured(4096)),
            'transforms_256.maray': encode((256, 256), scenes.transforms(256)),
            'test7_128.maray': encode((128, 128), [div(mul(nat(255), x()), nat(128))] * 3)}      # examples/test7.rs:5-7
    for name, data in want.items():
        assert open(os.path.join(scene_dir, name), 'rb').read() == data, name


def test_chess_board_reconstruction_lowers_and_matches_the_published_image(scene_dir):
    """examples/chess.rs rebuilt WITHOUT simplify/compress: 3.8 M tree nodes per channel collapse to a few
    thousand tape ops, and rows away from the two knife-edge rows equal images/chess.png exactly."""
    s = M.Scene.open(os.path.join(scene_dir, 'chess_board_1024.maray'))
    assert s.size == (1024, 1024) and s.node_count(0) > 3_000_000
    tape = s.lower()
    assert tape.info['n_pix_ops'] < 9000 and tape.info['sin_bounded'] == tape.info['sin_ops'] == 256
    png = np.asarray(Image.open(os.path.join(GOLDEN, 'chess.png')).convert('RGB'))
    for y0 in (100, 600, 800):
        got = tape_eval.cast_u8(tape_eval.render_rows(tape, 1024, y0, y0 + 3))
        assert np.array_equal(got, png[y0:y0 + 3])
    sdf = M.Scene.open(os.path.join(scene_dir, 'sdf_512.maray')).lower()
    img = tape_eval.cast_u8(tape_eval.render_rows(sdf, 512, 200, 312))
    assert set(np.unique(img)) == {0, 255}


def test_authored_scenes_are_what_the_examples_save(scene_dir):
    """maray_scenes with the examples' own `.simplify(mem).compress(mem)` (examples/chess.rs:43, examples/test.rs:21) through
    the library: a Let at the left of the top Mul like data/chess.maray (where src/wasm.rs:111-120 needs
    it), hundreds of variables, a file smaller than the stored one -- and the oracle's rows of it are images/chess.png,
    the knife-edge rows 512 / 704 aside.  The same example regenerated natively at 4096: pixel (4i,4j) is pixel (i,j)."""
    from marayb import decode
    from oracle_ffi import Scene as OScene
    data = open(os.path.join(scene_dir, 'chess_authored_1024.maray'), 'rb').read()
    assert len(data) < os.path.getsize(os.path.join(GOLDEN, 'chess.maray'))
    (w, h), color = decode(data)
    assert (w, h) == (1024, 1024) and color[0] == color[1] == color[2]
    assert color[0][0] == 'Mul' and color[0][1][0] == 'Let' and color[0][2] == ('Nat', 255) and len(color[0][1][1]) > 500
    png = np.asarray(Image.open(os.path.join(GOLDEN, 'chess.png')).convert('RGB'))
    o1 = OScene(data)
    for y0, y1 in ((100, 102), (510, 512), (513, 515), (600, 602), (703, 704), (705, 707), (818, 821)):
        got, _ = o1.render_rows(1024, 1024, y0, y1, want_f64=False)
        assert np.array_equal(got, png[y0:y1]), y0
    o4 = OScene(open(os.path.join(scene_dir, 'chess_authored_4096.maray'), 'rb').read())
    for r in (512, 600, 704):
        a, _ = o1.render_rows(1024, 1024, r, r + 1, want_f64=False)
        b, _ = o4.render_rows(4096, 4096, 4 * r, 4 * r + 1, want_f64=False)
        assert np.array_equal(a[0], b[0, ::4]), r
    sdf = open(os.path.join(scene_dir, 'sdf_512_authored.maray'), 'rb').read()
    assert len(sdf) < os.path.getsize(os.path.join(scene_dir, 'sdf_512.maray')) // 4
    a, _ = OScene(sdf).render_rows(512, 512, 200, 232, want_f64=False)
    b, _ = OScene(open(os.path.join(scene_dir, 'sdf_512.maray'), 'rb').read()).render_rows(512, 512, 200, 232, want_f64=False)
    assert (a != b).mean() < 0.002 and set(np.unique(a)) == {0, 255}        # simplify changes an association here and there, not the picture


@pytest.mark.gpu
def test_chess_board_reconstruction_full_image_on_gpu(scene_dir):
    s = M.Scene.open(os.path.join(scene_dir, 'chess_board_1024.maray'))
    png = np.asarray(Image.open(os.path.join(GOLDEN, 'chess.png')).convert('RGB'))
    img = M.gen_to_image(s)
    diff = np.any(img != png, axis=2)
    # a different association than the published picture's source: still only the two knife-edge rows differ
    assert diff.sum() <= 0.0004 * 1024 * 1024
    assert set(np.nonzero(diff.any(axis=1))[0].tolist()) <= {512, 704}


def test_transforming_and_curve_builders(scene_dir):
    """rotate / rotate_at / scale_at / translate, p2_cbez / p2_spiral / p2_subst, from_barycentric, p4_*, var / var_offset /
    var_range (src/lib.rs:738-850, 1031-1070, 1127-1151) in one picture, written by the C++ builders byte for byte as by the
    test builders (the test above), and meaning what they say when the oracle evaluates them: a point sent to barycentric
    coordinates and back is itself; a box scaled by (3/2, 1/2) and turned by 45 degrees about the centre, XOR a disk, has
    the disk's and the box's symmetry about the centre; the distance between the Bezier point and the spiral point is what
    numpy computes from the closed forms; `var("u")` is Rust's FNV-1a of the name with str's 0xff terminator."""
    from marayb import decode, var
    from oracle_ffi import Scene as OScene
    data = open(os.path.join(scene_dir, 'transforms_256.maray'), 'rb').read()
    (w, h), color = decode(data)
    assert (w, h) == (256, 256)
    u = 0xcbf29ce484222325
    for c in b'u\xff':
        u = ((u ^ c) * 0x100000001b3) % 2**64
    assert var('u') == ('Var', u) and color[2][0] == 'Let' and color[2][1][0][0] == (u + 7) % 2**64
    img8, img64 = OScene(data).render_rows(256, 256, 0, 256)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float64) / 256
    # B: x -> barycentric -> x, clamped to the unit interval, times 255
    assert np.allclose(img64[:, :, 2], np.clip(xx, 0, 1) * 255, atol=1e-9)
    # R: a set (0 or 255), symmetric under the half turn about the centre pixel centre-to-centre (x, y) -> (256 - x, 256 - y)
    r = img64[:, :, 0]
    assert set(np.unique(r)) == {0.0, 255.0} and 0.05 < (r == 255).mean() < 0.5
    assert (r[1:, 1:] != r[1:, 1:][::-1, ::-1]).mean() < 0.01               # (edges may fall on either side of a knife edge)
    # G: |cbez(t = x) - spiral(angle = tau y)| from the closed forms
    t = xx
    def lerp(a, b, t): return a + (b - a) * t
    def qb(a, b, c, t): return lerp(lerp(a, b, t), lerp(b, c, t), t)
    bx, by = lerp(qb(0, 1, 0, t), qb(1, 0, 1, t), t), lerp(qb(0, 0, 1, t), qb(0, 1, 1, t), t)
    ang = yy * 2 * np.pi
    sx, sy = np.cos(ang) * (ang / (2 * np.pi)), np.sin(ang) * (ang / (2 * np.pi))
    assert np.allclose(img64[:, :, 1], np.clip(np.hypot(bx - sx, by - sy), 0, 1) * 255, atol=1e-6)
