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


def test_png_reader_with_crafted_headers_under_asan_ubsan(tmp_path):
    """maray_png_read takes its sizes from the file (a texture named on the command line): an IHDR whose products
    overflow, or that promises more scanlines than its IDAT can inflate to, is an error code -- never an out-of-bounds
    access, an allocation the size of the lie, or an exception across the C boundary."""
    import struct
    import zlib

    import numpy as np
    csrc = os.path.join(ROOT, 'maray_amd', 'csrc')
    exe = str(tmp_path / 'png_asan')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-I' + csrc, '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'png_asan.cpp'),
                           os.path.join(csrc, 'png.cpp'), '-o', exe, '-lz'])

    def chunk(kind, data):
        return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data) & 0xFFFFFFFF)

    def png(w, h, depth, ctype, idat, interlace=0):
        return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, interlace)) +
                chunk(b'IDAT', idat) + chunk(b'IEND', b''))
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    raw = b''.join(b'\x00' + img[r].tobytes() for r in range(7))
    small = zlib.compress(b'\x00' * 16)
    files = {
        'good_rgb.png': png(5, 7, 8, 2, zlib.compress(raw)),
        'good_gray16.png': png(3, 2, 16, 0, zlib.compress(b''.join(b'\x01' + bytes(6) for _ in range(2)))),
        'huge_w.png': png(0xFFFFFFFF, 2, 8, 2, small),                       # stride * h wraps in 32 bits, not in 64: refused by size
        'huge_both.png': png(0xFFFFFFFF, 0xFFFFFFFF, 16, 6, small),          # (stride + 1) * h overflows 64 bits
        'wrap_to_small.png': png(0x55555556, 3, 8, 2, small),                # w * 3 wraps to 2 in 32 bits
        'lying_ihdr.png': png(4096, 4096, 8, 2, small),                      # plausible size, 16 bytes of pixels
        'million.png': png(1 << 20, 1 << 20, 8, 2, small),
        'truncated.png': png(5, 7, 8, 2, zlib.compress(raw))[:60],
        'bad_filter.png': png(5, 7, 8, 2, zlib.compress(b'\x09' + raw[1:])),
        'interlaced.png': png(5, 7, 8, 2, zlib.compress(raw), interlace=1),
        'zero.png': png(0, 0, 8, 2, small),
        'not_png.png': b'P6 5 7 255 ' + raw,
    }
    paths = []
    for name, data in files.items():
        p = str(tmp_path / name)
        open(p, 'wb').write(data)
        paths.append(p)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096',
               UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([exe] + paths, capture_output=True, text=True, env=env)
    assert out.returncode == 0 and 'png ok' in out.stdout, out.stdout[-2500:] + out.stderr[-3000:]


def test_image_readers_with_crafted_headers_under_asan_ubsan(tmp_path):
    """maray_image_read (BMP, PNM, TGA, QOI, farbfeld; tests/test_image_formats.py has the decodings) on headers that lie:
    sizes whose products overflow, pixel data shorter than promised, run-length packets past the end, palette indices past
    the palette, offsets past the file."""
    import struct

    import numpy as np
    from PIL import Image
    csrc = os.path.join(ROOT, 'maray_amd', 'csrc')
    exe = str(tmp_path / 'image_asan')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-I' + csrc, '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'image_asan.cpp'),
                           os.path.join(csrc, 'png.cpp'), os.path.join(csrc, 'image.cpp'), '-o', exe, '-lz'])
    rng = np.random.default_rng(3)
    im = Image.fromarray(rng.integers(0, 256, (9, 13, 3), dtype=np.uint8))
    paths = []
    for name in ('good.bmp', 'good.ppm', 'good.tga', 'good.png'):
        p = str(tmp_path / name)
        im.save(p)
        paths.append(p)
    im.convert('P', palette=Image.ADAPTIVE, colors=8).save(str(tmp_path / 'good_pal.bmp'))
    im.save(str(tmp_path / 'good_rle.tga'), compression='tga_rle')
    paths += [str(tmp_path / 'good_pal.bmp'), str(tmp_path / 'good_rle.tga')]

    def bmp(w, h, bpp, off=54, body=b'', hdr=40, ncol=0):
        return b'BM' + struct.pack('<IHHI', 0, 0, 0, off) + struct.pack('<IiiHHIIiiII', hdr, w, h, 1, bpp, 0, 0, 0, 0, ncol, 0) + body
    good_bmp = open(paths[0], 'rb').read()
    files = {
        'bmp_huge.bmp': bmp(1 << 20, 1 << 20, 24), 'bmp_wrap.bmp': bmp(0x7FFFFFFF, 3, 32), 'bmp_short.bmp': bmp(64, 64, 24, body=bytes(100)),
        'bmp_off.bmp': bmp(4, 4, 24, off=0xFFFFFFF0, body=bytes(64)), 'bmp_neg.bmp': bmp(-4, 4, 24, body=bytes(64)),
        'bmp_minh.bmp': bmp(4, -(1 << 31), 24, body=bytes(64)), 'good_pal3.bmp': bmp(4, 4, 8, off=54 + 12, body=bytes(12) + bytes([0, 1, 2, 200] * 4), ncol=3),      # an index past the palette reads as black
        'bmp_hdr.bmp': bmp(4, 4, 24, hdr=0xFFFFFF00, body=bytes(64)), 'bmp_trunc.bmp': good_bmp[:70],
        'pnm_huge.ppm': b'P6 1048576 1048576 255\n', 'pnm_short.ppm': b'P6 40 40 255\n' + bytes(100), 'pnm_big.ppm': b'P6 99999999999 2 255\n',
        'pnm_max.ppm': b'P6 2 2 70000\n' + bytes(24), 'pnm_plain.ppm': b'P3 4 4 255 1 2 3', 'pnm_bits.pbm': b'P4 17 3\n' + bytes(5),
        'tga_short.tga': bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 0, 64, 0, 24, 0]) + bytes(50),
        'tga_rle.tga': bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 0, 64, 0, 24, 0, 0xFF, 1, 2]),
        'good_map.tga': bytes([0, 1, 1, 0, 0, 2, 0, 24, 0, 0, 0, 0, 4, 0, 4, 0, 8, 0]) + bytes(6) + bytes([9] * 16),      # indices past the colour map read as black
        'tga_id.tga': bytes([255, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 4, 0, 24, 0]) + bytes(20),
        'qoi_short.qoi': b'qoif' + struct.pack('>IIBB', 64, 64, 3, 0) + bytes([0xFE, 1, 2]) + bytes(8),
        'qoi_huge.qoi': b'qoif' + struct.pack('>IIBB', 0xFFFFFFFF, 0xFFFFFFFF, 4, 0) + bytes(16),
        'ff_short.ff': b'farbfeld' + struct.pack('>II', 100, 100) + bytes(64), 'ff_huge.ff': b'farbfeld' + struct.pack('>II', 0xFFFFFFFF, 2),
    }
    # TIFF: Pillow's files in every compression, then truncated, with fields pointing outside the file, with sizes that lie
    for name, kw in (('good_raw.tif', {}), ('good_lzw.tif', {'compression': 'tiff_lzw', 'tiffinfo': {317: 2}}), ('good_zip.tif', {'compression': 'tiff_adobe_deflate'}),
                     ('good_pb.tif', {'compression': 'packbits'})):
        p = str(tmp_path / name)
        im.save(p, **kw)
        paths.append(p)
        t = open(p, 'rb').read()
        stem = name[5:-4]
        files['tif_%s_half.tif' % stem] = t[:len(t) // 2]
        files['tif_%s_tail.tif' % stem] = t[:-9]
        files['tif_%s_ifd.tif' % stem] = t[:4] + struct.pack('<I', len(t) - 3) + t[8:]
        k = t.find(struct.pack('<HHI', 256, 3, 1))                         # ImageWidth SHORT 1
        if k >= 0:
            files['tif_%s_wide.tif' % stem] = t[:k + 8] + struct.pack('<HH', 60000, 0) + t[k + 12:]
        k = t.find(struct.pack('<HH', 273, 4))                             # StripOffsets LONG
        if k >= 0:
            files['tif_%s_off.tif' % stem] = t[:k + 8] + struct.pack('<I', 0xFFFFFF00) + t[k + 12:]
        k = t.find(struct.pack('<HH', 279, 4))                             # StripByteCounts LONG
        if k >= 0:
            files['tif_%s_cnt.tif' % stem] = t[:k + 8] + struct.pack('<I', 0x7FFFFFFF) + t[k + 12:]
    files['tif_lzw_noise.tif'] = open(str(tmp_path / 'good_lzw.tif'), 'rb').read()[:8] + bytes(rng.integers(0, 256, 300, dtype=np.uint8)) + open(str(tmp_path / 'good_lzw.tif'), 'rb').read()[308:]
    # GIF: Pillow's files, and the same with bytes knocked out / the header lying / random LZW data
    for name, kw in (('good_a.gif', {}), ('good_i.gif', {'interlace': True})):
        p = str(tmp_path / name)
        im.convert('P', palette=Image.ADAPTIVE, colors=32).save(p, **kw)
        paths.append(p)
    big = Image.fromarray(rng.integers(0, 256, (120, 160), dtype=np.uint8), 'P')
    big.putpalette(bytes(rng.integers(0, 256, 768, dtype=np.uint8)))
    big.save(str(tmp_path / 'good_full_table.gif'))
    paths.append(str(tmp_path / 'good_full_table.gif'))
    g = open(str(tmp_path / 'good_full_table.gif'), 'rb').read()
    files.update({
        'gif_trunc1.gif': g[:20], 'gif_trunc2.gif': g[:len(g) // 2], 'gif_trunc3.gif': g[:800],
        'gif_screen.gif': g[:6] + struct.pack('<HH', 4, 4) + g[10:],                      # a frame larger than its logical screen
        'gif_huge.gif': g[:6] + struct.pack('<HH', 0xFFFF, 0xFFFF) + g[10:13] + b'\x3B',
        'gif_noise.gif': g[:13 + 768 + 10] + bytes([8]) + b''.join(bytes([255]) + bytes(rng.integers(0, 256, 255, dtype=np.uint8)) for _ in range(40)) + b'\0\x3B',
        'gif_min.gif': g[:13 + 768 + 10] + bytes([13]) + g[13 + 768 + 11:],
        'gif_ext.gif': g[:13 + 768] + b'\x21\xFE\xFF' + bytes(10),
    })
    # Headers that lie about sizes no file of a few hundred bytes can hold: nothing may be allocated,
    # reserved or zero-filled on a header's word.  These are also run by an unsanitized build under a 1 GiB address-space
    # limit below: an allocation the size of the lie would fail there (or, unlimited, zero-fill gigabytes).
    def tif(w, h, comp, spp, bits, data):
        ents = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, comp), (262, 3, 1, 2 if spp >= 3 else 1), (273, 4, 1, 8 + 2 + 12 * 10 + 4),
                (277, 3, 1, spp), (278, 4, 1, h), (279, 4, 1, len(data)), (284, 3, 1, 1)]
        ifd = struct.pack('<H', len(ents)) + b''.join(struct.pack('<HHII', t, ty, n, v) for t, ty, n, v in ents) + struct.pack('<I', 0)
        return b'II*\0' + struct.pack('<I', 8) + ifd + data
    gif_1x1 = b'\x2C' + struct.pack('<HHHHB', 0, 0, 1, 1, 0x80) + bytes(6) + bytes([2, 2, 0x4C, 0x01, 0]) + b'\x3B'
    lies = {
        'gif_blank_screen.gif': b'GIF89a' + struct.pack('<HHBBB', 0xFFFF, 0xFFFF, 0, 0, 0) + gif_1x1,            # 40 bytes, 12.9 GB of screen
        'gif_blank_screen2.gif': b'GIF89a' + struct.pack('<HHBBB', 20000, 20000, 0, 0, 0) + gif_1x1,
        'gif_frame_lies.gif': b'GIF89a' + struct.pack('<HHBBB', 9000, 9000, 0, 0, 0) + b'\x2C' + struct.pack('<HHHHB', 0, 0, 9000, 9000, 0x80) + bytes(6) + bytes([2, 2, 0x4C, 0x01, 0]) + b'\x3B',
        'tif_zip_lies.tif': tif(60000, 60000, 8, 8, 16, bytes(40)), 'tif_lzw_lies.tif': tif(60000, 60000, 5, 8, 16, bytes(40)),
        'tif_pb_lies.tif': tif(60000, 60000, 32773, 3, 8, bytes(40)), 'tif_raw_lies.tif': tif(60000, 60000, 1, 3, 8, bytes(40)),
        'tga_rle_lies.tga': bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 32, 0, 0xFF, 1, 2, 3, 4]),
        'tga_raw_lies.tga': bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 32, 0]) + bytes(64),
        'qoi_lies.qoi': b'qoif' + struct.pack('>IIBB', 40000, 40000, 3, 0) + bytes([0xFD] * 20) + bytes(8),
        'ff_lies.ff': b'farbfeld' + struct.pack('>II', 40000, 40000) + bytes(64),
        'bmp_lies.bmp': bmp(40000, 40000, 24, body=bytes(100)), 'pnm_lies.ppm': b'P6 40000 40000 255\n' + bytes(100), 'pnm_lies.pbm': b'P4 60000 60000\n' + bytes(100),
        'pnm_lies_plain.pgm': b'P2 40000 40000 255\n1 2 3',
    }
    files.update(lies)
    # Headers that lie about sizes no file of a few hundred bytes can hold: nothing may be allocated,
    # reserved or zero-filled on a header's word.  These are also run by an unsanitized build under a 1 GiB address-space
    # limit below: an allocation the size of the lie would fail there (or, unlimited, zero-fill gigabytes).
    def tif(w, h, comp, spp, bits, data):
        ents = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, comp), (262, 3, 1, 2 if spp >= 3 else 1), (273, 4, 1, 8 + 2 + 12 * 10 + 4),
                (277, 3, 1, spp), (278, 4, 1, h), (279, 4, 1, len(data)), (284, 3, 1, 1)]
        ifd = struct.pack('<H', len(ents)) + b''.join(struct.pack('<HHII', t, ty, n, v) for t, ty, n, v in ents) + struct.pack('<I', 0)
        return b'II*\0' + struct.pack('<I', 8) + ifd + data
    gif_1x1 = b'\x2C' + struct.pack('<HHHHB', 0, 0, 1, 1, 0x80) + bytes(6) + bytes([2, 2, 0x4C, 0x01, 0]) + b'\x3B'
    lies = {
        'gif_blank_screen.gif': b'GIF89a' + struct.pack('<HHBBB', 0xFFFF, 0xFFFF, 0, 0, 0) + gif_1x1,            # 40 bytes, 12.9 GB of screen
        'gif_blank_screen2.gif': b'GIF89a' + struct.pack('<HHBBB', 20000, 20000, 0, 0, 0) + gif_1x1,
        'gif_frame_lies.gif': b'GIF89a' + struct.pack('<HHBBB', 9000, 9000, 0, 0, 0) + b'\x2C' + struct.pack('<HHHHB', 0, 0, 9000, 9000, 0x80) + bytes(6) + bytes([2, 2, 0x4C, 0x01, 0]) + b'\x3B',
        'tif_zip_lies.tif': tif(60000, 60000, 8, 8, 16, bytes(40)), 'tif_lzw_lies.tif': tif(60000, 60000, 5, 8, 16, bytes(40)),
        'tif_pb_lies.tif': tif(60000, 60000, 32773, 3, 8, bytes(40)), 'tif_raw_lies.tif': tif(60000, 60000, 1, 3, 8, bytes(40)),
        'tga_rle_lies.tga': bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 32, 0, 0xFF, 1, 2, 3, 4]),
        'tga_raw_lies.tga': bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 32, 0]) + bytes(64),
        'qoi_lies.qoi': b'qoif' + struct.pack('>IIBB', 40000, 40000, 3, 0) + bytes([0xFD] * 20) + bytes(8),
        'ff_lies.ff': b'farbfeld' + struct.pack('>II', 40000, 40000) + bytes(64),
        'bmp_lies.bmp': bmp(40000, 40000, 24, body=bytes(100)), 'pnm_lies.ppm': b'P6 40000 40000 255\n' + bytes(100), 'pnm_lies.pbm': b'P4 60000 60000\n' + bytes(100),
        'pnm_lies_plain.pgm': b'P2 40000 40000 255\n1 2 3',
    }
    files.update(lies)
    for name, data in files.items():
        p = str(tmp_path / name)
        open(p, 'wb').write(data)
        paths.append(p)
    plain = str(tmp_path / 'image_plain')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-I' + csrc, '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'image_asan.cpp'),
                           os.path.join(csrc, 'png.cpp'), os.path.join(csrc, 'image.cpp'), '-o', plain, '-lz'])
    import resource

    def limit():
        resource.setrlimit(resource.RLIMIT_AS, (1 << 30, 1 << 30))
    out = subprocess.run([plain] + [str(tmp_path / n) for n in lies] + [paths[0]], capture_output=True, text=True, preexec_fn=limit)
    assert out.returncode == 0 and 'images ok' in out.stdout and 'out of memory' not in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('rc 0') == 1                                    # only the good file decodes
    plain = str(tmp_path / 'image_plain')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-I' + csrc, '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'image_asan.cpp'),
                           os.path.join(csrc, 'png.cpp'), os.path.join(csrc, 'image.cpp'), '-o', plain, '-lz'])
    import resource

    def limit():
        resource.setrlimit(resource.RLIMIT_AS, (1 << 30, 1 << 30))
    out = subprocess.run([plain] + [str(tmp_path / n) for n in lies] + [paths[0]], capture_output=True, text=True, preexec_fn=limit)
    assert out.returncode == 0 and 'images ok' in out.stdout and 'out of memory' not in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('rc 0') == 1                                    # only the good file decodes
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096',
               UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([exe] + paths, capture_output=True, text=True, env=env)
    assert out.returncode == 0 and 'images ok' in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
