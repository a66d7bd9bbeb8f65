#!/usr/bin/env python3
"""tools/gen_tape_hashes.py — writes tests/golden/tape_hashes.json: SHA-256 of the tapes (constants, ROW ops, PIXEL ops, counts) the
lowering makes of a fixed set of scenes.  tests/test_tape_stability.py compares: a change to the lowering that is meant to keep
its output (a faster table, another walk order) must leave every hash alone — the specialised kernels' code keys, and with them
the committed profiles' `matches_this_build`, follow from the tapes.  A change that is MEANT to alter tapes regenerates the file
(and re-collects the profiles)."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import maray_amd as M  # noqa: E402


def tape_hash(t):
    p = t.program
    h = hashlib.sha256()
    h.update(C.string_at(p.consts, p.n_consts * 8))
    h.update(C.string_at(p.row_ops, p.n_row_ops * 8))
    h.update(C.string_at(p.pix_ops, p.n_pix_ops * 8))
    h.update(repr((p.n_yvals, p.n_row_slots, p.n_pix_slots, p.n_app)).encode())
    return h.hexdigest()[:24]


def scenes_to_hash():
    import fuzz_scenes
    import scenes
    from marayb import encode
    data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
    for sc in [(1, 1), (4, 4), (4, 16), (16, 16)]:
        for kw in ({}, {'skips': False}):
            s = M.Scene(data)
            s.rescale(*sc)
            yield 'chess x%d y%d %s' % (sc[0], sc[1], 'plain' if kw else 'guarded'), s, kw
    for name, e in [('all_ops 512', scenes.all_ops(512, 512)), ('radial', scenes.radial_gradient()), ('textured 512', scenes.textured(512))]:
        yield name, M.Scene(encode((512, 512), e)), {}
    for seed in range(6):
        yield 'polygon soup %d' % seed, M.Scene(encode((512, 384), fuzz_scenes.polygon_soup(seed, 20 + 7 * seed, 512, 384))), {}
        yield 'curved soup %d' % seed, M.Scene(encode((512, 384), fuzz_scenes.curved_soup(seed, 10 + 5 * seed, 512, 384))), {}
        yield 'product soup %d' % seed, M.Scene(encode((512, 384), fuzz_scenes.product_soup(seed, 10 + 5 * seed, 512, 384))), {}
        yield 'random scene %d' % seed, M.Scene(encode((512, 384), fuzz_scenes.scene(seed, depth=4 + seed))), {}


def all_hashes():
    return {name: tape_hash(s.lower(**kw)) for name, s, kw in scenes_to_hash()}


if __name__ == '__main__':
    out = os.path.join(ROOT, 'tests', 'golden', 'tape_hashes.json')
    json.dump(all_hashes(), open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out)
