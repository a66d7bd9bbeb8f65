"""The hiprtc back-end generates and compiles gfx950 code without a GPU."""
import os
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


def test_chess_uses_the_pure_bounded_step_sin(chess_bytes, monkeypatch):
    tape = M.Scene(chess_bytes).lower()
    assert tape.info['sin_ops'] == 256 and tape.info['sin_bounded'] == 256
    L = M.lib()
    L.maray_jit_source.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]

    def source():
        src = C.c_void_p()
        assert L.maray_jit_source(C.byref(tape.program), C.byref(src)) == 0
        text = C.string_at(src).decode()
        L.maray_free(src)
        return text
    import re
    text = source()
    assert text.count('mr_stepsin_bounded_mk(') == 256 and 'mr_stepsin_fast(' not in text
    assert text.count('const mr_mask ') > 2000       # half of chess is boolean algebra on lane masks (SGPR pairs)
    assert text.count('mr_mask bv') > 2800 and ' bool bv' not in text and ' bool v' not in text
    # the tile with no guard bit set is four pixels per lane of a colour the emitter folded (0.0 * 255: no load, no
    # multiply); y values that are booleans are read as masks
    assert 'mr_d o0 = 0.0' in text and '* mr_kc[0]' not in text.split('mr_d o0 = 0.0')[1].split('mr_u3')[0] and text.count('mr_ym(yw, ') >= 64 and 'mr_min(' not in text and 'mr_max(' not in text
    # The scene's OR tree of 128 guarded shapes is a reduction: per guard word the set bits of the rectangle at hand are
    # taken lowest first and reach their shape through a branch table; no bit test of the tree is left in the busy variant,
    # group guards and "all lanes covered" regions included
    assert text.count('asm goto(') == 3 and text.count('__builtin_ctzll(mr_rm)') == 3
    assert len(set(re.findall(r'(mr_rl\d+_0_\d+): \{', text))) == 128 and text.count('mr_racc1_0 |= ') == 128
    assert not re.findall(r'\(unsigned\)\(?gq\d(?: >> 32\))? & 0x[0-9a-f]+u\)\) != 0u', text)
    # MARAY_JIT_REDUCE=0: the tree as written -- a group's guard has no bit of its own: its test is a mask over its members' bits
    monkeypatch.setenv('MARAY_JIT_REDUCE', '0')
    text = source()
    assert 'asm goto(' not in text
    assert len(re.findall(r'\(unsigned\)\(?gq\d(?: >> 32\))? & 0x[0-9a-f]+u\)\) != 0u', text)) >= 160


def test_row_section_is_cut_into_chunks(chess_bytes):
    """maray_jit_rows evaluates independent jobs side by side (blockIdx.y).  The first k are chunks of the ROW section:
    every y value the pixel kernel reads as an operand is written by exactly one of them.  The rest are guard words:
    the guards are evaluated per 256-pixel tile (XMIN / XMAX = the tile's ends) and packed 64 to a word.  It builds."""
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
    rows_src, guards_src = text.split('// guards: (row group, tile)')
    assert 2 <= k.value <= 64 and rows_src.count('    case ') == k.value
    # a chunk's values go to LDS ([value - first][row]) and leave as rows of the table: every operand y value once
    first = [int(m) for m in re.findall(r'mr_k0 = (\d+)u; mr_kn = \d+u;', rows_src)]
    count = [int(m) for m in re.findall(r'mr_k0 = \d+u; mr_kn = (\d+)u;', rows_src)]
    assert len(first) == k.value and first[0] == 0 and all(first[i] + count[i] == first[i + 1] for i in range(k.value - 1)) and max(count) <= 16
    n_num = first[-1] + count[-1]
    assert 0 < n_num < tape.info['n_yvals'] and len(re.findall(r'    ys\[\d+u \+ mr_lane\] = ', rows_src)) == n_num
    n_guards = tape.info['n_yvals'] - n_num
    # guards that are the OR of other guards (a group of shapes) have no job; the others, 8 bits = one byte per job
    n_bits = len(re.findall(r'gacc \|= ', guards_src))
    assert 0 < n_bits < n_guards and n_bits >= 128 and 'ys[' not in guards_src.split('switch (')[1]
    assert guards_src.count('    case ') == (n_bits + 7) // 8
    assert 'XMIN' in guards_src and 'XMIN' not in rows_src.split('switch (')[1]      # only guards depend on the span
    build(tape)        # compiles the ROW kernel as well


def test_random_scenes_build():
    """The generated sources of random scenes compile (both kernels).  Seeds 1270..1289 include one where a lane mask
    defined outside a region was first materialised as an f64 inside it and used again after it (a name out of scope)."""
    from test_fuzz import lowered
    built = 0
    for seed in list(range(0, 60, 3)) + list(range(1270, 1290)):
        _, tape = lowered(seed, 2 if seed % 3 == 0 else 0)
        if tape is None:
            continue
        build(tape)
        built += 1
    assert built >= 35


def test_ops_on_literal_operands_build_in_every_form(monkeypatch):
    """In the variant for tiles without a guard bit the value of every guarded shape is the literal 0.0: ops meet operands
    that are numbers, not vector values -- texture coordinates too (seed 6462 of round 4's second GPU sweep: `mr_texel(t, tex,
    0.0, yv[2])` resolved to the one-pixel form inside the four-pixel variant and the kernel did not compile).  The crafted
    scene does that to every kind of op; both builds with one and with two rows per wavefront."""
    import re
    from test_fuzz import lowered
    tape = M.Scene(encode((512, 256), scenes.ops_on_a_guarded_mask(512, 256))).lower()
    text, _ = build(tape)
    sky = text[text.index('mr_d o0 = 0.0'):]
    sky = sky[:sky.index('mr_u3')]
    assert 'mr_texel(mr_t0, tex, mr_d(0.0), mr_d(0.0))' in sky and len(re.findall(r'mr_texel\(', sky)) == 4
    assert 'mr_stepsin_bounded_m(0.0)' in sky and 'mr_ln(1.0)' in sky
    _, tape2 = lowered(6462, 2)
    build(tape2)
    _, tape3 = lowered(8157, 2)      # its texture lookup is op 4: the texel's variable was named mr_tx4, which is a type
    build(tape3)
    monkeypatch.setenv('MARAY_JIT_ROWS2', '1')
    for t in (tape, tape2, tape3):
        build(t)


def test_code_key_is_remembered_under_the_programs_name(chess_bytes, tmp_path, monkeypatch):
    """The code key hashes the generated sources (0.1 s for chess); it is remembered under a hash of the program, the
    library's build and the MARAY_JIT_* knobs, in the process and in <cache>/<name>.key.  A knob that changes the source
    changes the key, one that does not (a launch-time knob) maps another name onto the same key; a cache directory that
    cannot be written is not an error."""
    import os
    import time
    monkeypatch.setenv('MARAY_CACHE_DIR', str(tmp_path))
    tape = M.Scene(chess_bytes).lower()
    t0 = time.perf_counter()
    key = tape.jit_code_key
    t_first = time.perf_counter() - t0
    names = sorted(os.listdir(tmp_path))
    assert len(key) == 32 and len(names) == 1 and names[0].endswith('.key') and open(tmp_path / names[0]).read() == key
    t0 = time.perf_counter()
    assert tape.jit_code_key == key
    assert time.perf_counter() - t0 < t_first / 4                       # no source generation the second time
    assert not tape.jit_code_cached                                     # nothing was built: there are no code objects yet
    monkeypatch.setenv('MARAY_JIT_TILES', '3')                          # launch-time knob: same sources
    assert tape.jit_code_key == key and len(os.listdir(tmp_path)) == 2
    monkeypatch.setenv('MARAY_JIT_GUARD_W', '256')                      # changes the generated kernels
    other = tape.jit_code_key
    assert other != key and len(os.listdir(tmp_path)) == 3
    monkeypatch.delenv('MARAY_JIT_GUARD_W')
    # the rectangle's height is a launch parameter that a code object carries with it (the cache file's header): it names
    # the object too, or a cached default would be launched with its own height whatever the knob says
    monkeypatch.setenv('MARAY_JIT_GUARD_H', '8')
    assert tape.jit_code_key not in (key, other)
    monkeypatch.delenv('MARAY_JIT_GUARD_H')
    monkeypatch.delenv('MARAY_JIT_TILES')
    # a name file that is not a key is ignored and rewritten
    open(tmp_path / names[0], 'w').write('not a key')
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]; import maray_amd as M; "
            "print(M.Scene(open(%r, 'rb').read()).lower().jit_code_key)" % (os.path.dirname(here), here, os.path.join(here, 'golden', 'chess.maray')))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ), timeout=300)
    assert out.stdout.strip() == key, out.stderr[-500:]
    assert open(tmp_path / names[0]).read() == key
    monkeypatch.setenv('MARAY_CACHE_DIR', '/proc/no/such/dir')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ), timeout=300)
    assert out.stdout.strip() == key, out.stderr[-500:]


def test_helper_processes_build_what_the_process_itself_builds(monkeypatch, tmp_path):
    """The two modules of a program are built by two maray_jitc processes side by side (hiprtc serialises compiles inside
    a process); MARAY_JIT_HELPER=0 compiles in-process.  Same compiler, same options: the same code objects.  A source
    that does not compile comes back as MARAY_E_HIP with the compiler's log either way."""
    import hashlib
    monkeypatch.setenv('AMD_COMGR_CACHE', '0')
    tape = M.Scene(encode((64, 64), scenes.all_ops(64, 64))).lower()
    digests = []
    for helper in ('1', '0'):
        monkeypatch.setenv('MARAY_JIT_HELPER', helper)
        monkeypatch.setenv('MARAY_CACHE_DIR', str(tmp_path / ('cache' + helper)))
        _, blob = build(tape)
        assert tape.jit_code_cached
        digests.append(hashlib.sha256(blob).hexdigest())
    assert digests[0] == digests[1]
    monkeypatch.setenv('MARAY_JIT_OPT', '-Onotanoption')
    for helper in ('1', '0'):
        monkeypatch.setenv('MARAY_JIT_HELPER', helper)
        monkeypatch.setenv('MARAY_CACHE_DIR', str(tmp_path / ('bad' + helper)))
        L = M.lib()
        code, n = C.c_void_p(), C.c_size_t()
        assert L.maray_jit_build(C.byref(tape.program), C.byref(code), C.byref(n)) == -9          # MARAY_E_HIP
        assert b'notanoption' in L.maray_last_error()


def test_a_helper_that_dies_compiling_is_an_error_not_a_retry_in_process(monkeypatch, tmp_path):
    """An LLVM abort inside hiprtc is what the helper keeps out of the caller (gpu_tests5.log of round 2: `Fatal Python
    error: Aborted` inside M.Context).  MARAY_JITC_TEST_ABORT=1 makes the helper end that way: the build comes back as
    MARAY_E_HIP naming the signal and the kept source, this process lives on, nothing lands in the cache -- and the same
    source is NOT compiled in-process (it would have built fine there, so success would mean exactly that retry)."""
    import glob
    monkeypatch.setenv('AMD_COMGR_CACHE', '0')
    monkeypatch.setenv('MARAY_JIT_HELPER', '1')
    monkeypatch.setenv('MARAY_JITC_TEST_ABORT', '1')
    monkeypatch.setenv('MARAY_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('TMPDIR', str(tmp_path))
    tape = M.Scene(encode((80, 48), scenes.all_ops(80, 48))).lower()      # a program no other test of this process has built
    L = M.lib()
    code, n = C.c_void_p(), C.c_size_t()
    assert L.maray_jit_build(C.byref(tape.program), C.byref(code), C.byref(n)) == -9          # MARAY_E_HIP
    msg = L.maray_last_error().decode()
    assert 'compiler aborted' in msg and 'signal 6' in msg and 'source kept in' in msg, msg
    kept = glob.glob(str(tmp_path / 'maray_jit_*_pix_*' / 'kernel.hip'))
    assert kept and 'maray_jit_pixels' in open(kept[0]).read()
    assert (os.stat(os.path.dirname(kept[0])).st_mode & 0o777) == 0o700                        # a directory of its own
    assert not tape.jit_code_cached
    # the process survived and the library still works: without the knob the same program builds
    monkeypatch.delenv('MARAY_JITC_TEST_ABORT')
    _, blob = build(tape)
    assert blob[:4] == b'\x7fELF'


def test_a_damaged_cache_file_is_a_miss(tmp_path):
    """The launch geometry of cached kernels comes from the .mrco header: a damaged header -- even one whose checksum has
    been made to fit -- must read as a miss and be rebuilt, not divide by a zero guard width or under-size a guard table
    (ADVICE round 3).  Each step in a process of its own: the process's own table would hide the file."""
    import glob
    import struct
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, ctypes as C\nsys.path[:0] = [%r, %r]\nimport maray_amd as M, scenes\nfrom marayb import encode\n"
            "t = M.Scene(encode((72, 40), scenes.all_ops(72, 40))).lower()\nL = M.lib()\n"
            "L.maray_jit_build.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]\n"
            "c, n = C.c_void_p(), C.c_size_t()\nrc = L.maray_jit_build(C.byref(t.program), C.byref(c), C.byref(n))\n"
            "print(rc, n.value)\n" % (root, os.path.join(root, 'tests')))
    env = dict(os.environ, MARAY_CACHE_DIR=str(tmp_path), AMD_COMGR_CACHE='0')

    def run():
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0 and out.stdout.split()[0] == '0', out.stdout + out.stderr[-800:]
        return int(out.stdout.split()[1])
    n0 = run()
    (path,) = glob.glob(str(tmp_path / '*.mrco'))
    good = open(path, 'rb').read()
    hdr = list(struct.unpack('<10I', good[:40]))
    assert hdr[0] == 0x3463726d and hdr[4] == n0 and hdr[7] in (64, 128, 256) and hdr[9] in (0, 1) and len(good) == 40 + hdr[4] + hdr[5] + 8

    def fnv(data, h=0xcbf29ce484222325):
        for b in data:
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h
    assert struct.unpack('<Q', good[-8:])[0] == fnv(good[40 + hdr[4]:-8], fnv(good[40:40 + hdr[4]], fnv(good[:40])))     # the sum covers the header
    for field, value in ((7, 0), (7, 96), (8, 24), (8, 0), (6, 5000), (3, 3), (1, 0), (9, 2)):
        bad = list(hdr)
        bad[field] = value
        head = struct.pack('<10I', *bad)
        body = good[40:-8]
        forged = head + body + struct.pack('<Q', fnv(good[40 + hdr[4]:-8], fnv(good[40:40 + hdr[4]], fnv(head))))   # checksum made to fit
        for blob in (forged, head + body + good[-8:]):
            open(path, 'wb').write(blob)
            os.utime(path, (1, 1))
            assert run() == n0                                   # rebuilt, same kernels
            assert open(path, 'rb').read() == good               # ... and the file is whole again


def _sources(tape):
    L = M.lib()
    L.maray_jit_source.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.maray_jit_source_rows.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    out = []
    for fn, extra in ((L.maray_jit_source, ()), (L.maray_jit_source_rows, (C.byref(C.c_uint32()),))):
        src = C.c_void_p()
        assert fn(C.byref(tape.program), C.byref(src), *extra) == 0, L.maray_last_error()
        out.append(C.string_at(src).decode())
        L.maray_free(src)
    return out


def test_guard_rectangles_are_the_back_ends_choice(chess_bytes, monkeypatch):
    """Guards are bounded over rectangles of 64 pixels x 32 rows by default (jit_guard_geom): the ROW kernel's guard items
    span 64 pixels, the PIXEL kernel tests a tile's twelve words with one ballot and takes a pass's three by v_readlane.
    A program with more than 12 guard words holds a tile's words one per lane and still gets narrow rectangles when they
    fit a wavefront's 64 lanes."""
    tape = M.Scene(chess_bytes).lower()
    pix, rows = _sources(tape)
    assert 'tile * 64u' in rows and 'tile * 256u' not in rows
    assert 'mr_gnz' in pix and '(t * 4u + (e >> 0u)) * 3u' in pix
    for env, width, marker in (({'MARAY_JIT_GUARD_W': '128'}, 128, '(t * 2u + (e >> 1u)) * 3u'), ({'MARAY_JIT_GUARD_W': '256'}, 256, 't * 3u + 0u')):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pix, rows = _sources(tape)
        for k in env:
            monkeypatch.delenv(k)
        assert 'tile * %du' % width in rows, env
        assert marker in pix, env
    # 1,000 triangles: 16 guard words per rectangle, four rectangles per tile = the 64 lanes of a wavefront
    import fuzz_scenes
    soup = M.Scene(encode((4096, 4096), fuzz_scenes.polygon_soup(7, 1000, 4096, 4096, mixed=False))).lower()
    pix, rows = _sources(soup)
    assert 'tile * 64u' in rows and 'mr_lane < 64u ?' in pix and 'mr_gsub' in pix
    # ... and its OR tree of ~990 shapes is a reduction: a branch table per guard word, words without a bit skipped by one ballot
    # per pass.  (Shapes that differ in constants and y values alone sharing ONE body that reads both from tables was built
    # and measured: a third of the code, and 0.203 against 0.138 ms per frame -- the dependent loads cost more than the
    # instruction fetches save.  Not kept.)
    import re
    assert pix.count('asm goto(') == 16 and 'mr_gnzp' in pix and len(re.findall(r'mr_rl\d+_\d+_\d+: \{', pix)) > 900
