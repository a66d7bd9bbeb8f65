"""The lowering's output on a fixed set of scenes, hash by hash (tests/golden/tape_hashes.json, tools/gen_tape_hashes.py).  The
specialised kernels are generated from the tape, their code key is the hash of the generated sources, and the committed
profiles name the code keys they were taken on (`roofline.traffic_profile.matches_this_build` in bench.py's line): a change to
the lowering that is not meant to change its output must leave these alone."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_the_tapes_of_the_fixed_scenes_are_what_they_were():
    import gen_tape_hashes
    want = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tape_hashes.json')))
    got = gen_tape_hashes.all_hashes()
    assert sorted(got) == sorted(want)
    changed = [k for k in want if got[k] != want[k]]
    assert not changed, 'tapes changed (regenerate with tools/gen_tape_hashes.py if that was meant, and re-collect the profiles): %s' % changed
