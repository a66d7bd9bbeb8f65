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
from marayb import encode


@pytest.fixture(scope='module')
def scene_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp('scenes'))
    exe = os.path.join(os.path.dirname(M.lib_path()), 'maray_scenes')
    subprocess.check_call([exe, d])
    return d


def test_cpp_builders_write_the_same_files_as_the_test_builders(scene_dir):
    want = {'radial_1024.maray': encode((1024, 1024), scenes.radial_gradient()),
            'allops_4096.maray': encode((4096, 4096), scenes.all_ops(4096, 4096)),
            'textured_4096.maray': encode((4096, 4096), scenes.textured(4096))}
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
