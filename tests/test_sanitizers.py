"""Host logic of the product (bincode reader, fix_color, lowering, program validator) under AddressSanitizer and
UBSan — CPU build only (GPU sanitizers are not available on this pool)."""
import os
import subprocess

import scenes
from conftest import GOLDEN, ROOT
from fuzz_scenes import polygon_soup, scene
from marayb import encode


def test_reader_and_lowering_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, 'maray_amd', 'csrc')
    exe = str(tmp_path / 'lower_asan')
    # api.cpp carries validate_program next to HIP-dependent code; compile only that function's translation unit
    # by defining the HIP-free subset through the preprocessor is not possible, so extract it via a tiny shim:
    shim = str(tmp_path / 'validate_shim.cpp')
    src = open(os.path.join(csrc, 'api.cpp')).read()
    a = src.index('void validate_program(const maray_program &p)')
    b = src.index('}   // namespace maray', a)
    with open(shim, 'w') as f:
        f.write('#include <string>\n#include <vector>\n#include "expr.hpp"\n#include "lower.hpp"\n#include "maray_hip.h"\nnamespace maray {\nuint32_t numeric_yvals(const maray_program &P);\n'
                + src[a:b] + '}\n')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-ffp-contract=off', '-I' + csrc, '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'native', 'lower_asan.cpp'), os.path.join(csrc, 'scene.cpp'),
                           os.path.join(csrc, 'simplify.cpp'), os.path.join(csrc, 'lower.cpp'), os.path.join(csrc, 'row_split.cpp'), shim, '-o', exe, '-lpthread'])
    files = [os.path.join(GOLDEN, 'chess.maray')]
    for k, data in enumerate([encode((64, 64), scenes.all_ops(64, 64)), encode((64, 64), scenes.textured(64))] +
                             [encode((83, 9), scene(seed, n_tex=2 if seed % 3 == 0 else 0)) for seed in range(40)] +
                             [encode((512, 128), polygon_soup(7, 40, 512, 128, mixed=kind)) for kind in (True, False, 'colours')] +
                             [encode((192, 24), scenes.shapes_through_inf_and_nan())]):
        p = str(tmp_path / ('s%d.maray' % k))
        open(p, 'wb').write(data)
        files.append(p)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([exe] + files, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert 'lowered' in out.stdout
