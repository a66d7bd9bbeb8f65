"""Texture files: `image::open(file).to_rgb8()` (/root/reference/examples/maray.rs:58-65) takes whatever the `image` crate
knows; maray_image_read takes PNG (every colour type and depth, interlaced too), BMP, PNM, TGA, QOI and farbfeld, told from
the file's first bytes, and names the codecs it does not restate.  Reference decodings: Pillow, and the formats' own
definitions where Pillow has no writer."""
import os
import struct

import numpy as np
import pytest
from PIL import Image

import maray_amd as M


@pytest.fixture(scope='module')
def pic():
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    a[5:20, 7:30] = (200, 30, 90)           # runs, for the run-length formats
    a[25:, :10] = 0
    return a


def test_png_every_colour_type_and_interlace(tmp_path, pic):
    im = Image.fromarray(pic)
    cases = {'rgb': im, 'rgba': im.convert('RGBA'), 'grey': im.convert('L'), 'greya': im.convert('LA'),
             'pal': im.convert('P', palette=Image.ADAPTIVE, colors=64), 'bw': im.convert('1')}
    for name, x in cases.items():
        for interlace in (False, True):
            p = str(tmp_path / ('%s_%d.png' % (name, interlace)))
            if interlace:       # Pillow cannot write Adam7: interlace the file by hand from its pixels
                _write_adam7(p, x)
            else:
                x.save(p)
            assert np.array_equal(M.image_read(p), np.asarray(Image.open(p).convert('RGB'))), (name, interlace)
            assert np.array_equal(M.png_read(p), M.image_read(p))
    p = str(tmp_path / 'g16.png')
    Image.fromarray((np.arange(37 * 53, dtype=np.uint16).reshape(37, 53) * 33)).save(p)
    v = (np.arange(37 * 53, dtype=np.uint32).reshape(37, 53) * 33) & 0xFFFF
    assert np.array_equal(M.image_read(p)[:, :, 0], ((v * 255 + 32767) // 65535).astype(np.uint8))      # image's 16 -> 8 rounding


def _write_adam7(path, im):
    """The same pixels as an interlaced PNG (RFC 2083 section 2.6), filter 0 on every scanline."""
    import zlib
    mode = im.mode
    a = np.asarray(im if mode != '1' else im.convert('L').point(lambda v: 1 if v else 0, 'L'))
    ctype = {'RGB': 2, 'RGBA': 6, 'L': 0, 'LA': 4, 'P': 3, '1': 0}[mode]
    depth = 1 if mode == '1' else 8
    if a.ndim == 2:
        a = a[:, :, None]
    raw = b''
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            if depth == 1:
                bits = np.packbits(row[:, 0].astype(np.uint8))
                raw += b'\x00' + bits.tobytes()
            else:
                raw += b'\x00' + row.astype(np.uint8).tobytes()

    def chunk(kind, data):
        return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data) & 0xFFFFFFFF)
    out = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', im.size[0], im.size[1], depth, ctype, 0, 0, 1))
    if mode == 'P':
        out += chunk(b'PLTE', bytes(im.getpalette()[:768]))
    out += chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b'')
    open(path, 'wb').write(out)
    # Pillow reads interlaced files: the hand-made file is a PNG as others see it
    assert np.array_equal(np.asarray(Image.open(path).convert('RGB')), np.asarray(im.convert('RGB')))


def test_bmp_pnm_tga(tmp_path, pic):
    im = Image.fromarray(pic)
    files = {'a.bmp': im, 'b.bmp': im.convert('P', palette=Image.ADAPTIVE, colors=32), 'c.bmp': im.convert('RGBA'), 'd.bmp': im.convert('L'),
             'a.ppm': im, 'b.pgm': im.convert('L'), 'c.pbm': im.convert('1'),
             'a.tga': im, 'b.tga': im.convert('RGBA'), 'c.tga': im.convert('L'), 'd.tga': im.convert('P', palette=Image.ADAPTIVE, colors=32)}
    for name, x in files.items():
        p = str(tmp_path / name)
        x.save(p)
        assert np.array_equal(M.image_read(p), np.asarray(Image.open(p).convert('RGB'))), name
    p = str(tmp_path / 'rle.tga')
    im.save(p, compression='tga_rle')
    assert np.array_equal(M.image_read(p), pic)
    # plain PNM, comments, 16-bit maxval
    p = str(tmp_path / 'plain.ppm')
    open(p, 'w').write('P3\n# a comment\n2 2\n# another\n1023\n0 511 1023  1023 0 0\n5 5 5  1000 1000 1000\n')
    want = np.array([[[0, 511, 1023], [1023, 0, 0]], [[5, 5, 5], [1000, 1000, 1000]]], dtype=np.uint64)
    v16 = (want * 65535 + 511) // 1023
    assert np.array_equal(M.image_read(p), ((v16 * 255 + 32767) // 65535).astype(np.uint8))
    p = str(tmp_path / 'plain.pbm')
    open(p, 'w').write('P1 3 2\n101\n0 1 0\n')
    assert np.array_equal(M.image_read(p)[:, :, 0], np.array([[0, 255, 0], [255, 0, 255]], dtype=np.uint8))
    # top-down BMP (negative height)
    p = str(tmp_path / 'topdown.bmp')
    row = (53 * 3 + 3) // 4 * 4
    body = b''.join(pic[y, :, ::-1].tobytes() + bytes(row - 53 * 3) for y in range(37))
    open(p, 'wb').write(b'BM' + struct.pack('<IHHI', 54 + len(body), 0, 0, 54) + struct.pack('<IiiHHIIiiII', 40, 53, -37, 1, 24, 0, len(body), 0, 0, 0, 0) + body)
    assert np.array_equal(M.image_read(p), pic)


def test_qoi_and_farbfeld(tmp_path, pic):
    def qoi_encode(a):          # qoiformat.org reference encoder, RGB
        h, w, _ = a.shape
        out = bytearray(b'qoif' + struct.pack('>IIBB', w, h, 3, 0))
        idx = [(0, 0, 0, 0)] * 64
        prev = (0, 0, 0, 255)
        run = 0
        px_list = [tuple(int(v) for v in p) + (255,) for p in a.reshape(-1, 3)]
        for i, px in enumerate(px_list):
            if px == prev:
                run += 1
                if run == 62 or i == len(px_list) - 1:
                    out.append(0xC0 | (run - 1)); run = 0
                continue
            if run:
                out.append(0xC0 | (run - 1)); run = 0
            k = (px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64
            if idx[k] == px:
                out.append(k)
            else:
                idx[k] = px
                d = [((px[c] - prev[c] + 128) & 255) - 128 for c in range(3)]
                if all(-2 <= v <= 1 for v in d):
                    out.append(0x40 | (d[0] + 2) << 4 | (d[1] + 2) << 2 | (d[2] + 2))
                elif -32 <= d[1] <= 31 and -8 <= d[0] - d[1] <= 7 and -8 <= d[2] - d[1] <= 7:
                    out.append(0x80 | (d[1] + 32)); out.append((d[0] - d[1] + 8) << 4 | (d[2] - d[1] + 8))
                else:
                    out += bytes([0xFE, px[0], px[1], px[2]])
            prev = px
        return bytes(out) + bytes(7) + b'\x01'
    p = str(tmp_path / 'a.qoi')
    open(p, 'wb').write(qoi_encode(pic))
    assert np.array_equal(M.image_read(p), pic)
    p = str(tmp_path / 'a.ff')
    v16 = pic.astype(np.uint16) * 257
    rgba = np.concatenate([v16, np.full(pic.shape[:2] + (1,), 65535, np.uint16)], axis=2)
    open(p, 'wb').write(b'farbfeld' + struct.pack('>II', 53, 37) + rgba.astype('>u2').tobytes())
    assert np.array_equal(M.image_read(p), pic)


def test_formats_not_restated_and_broken_files_are_errors(tmp_path, pic):
    Image.fromarray(pic).save(str(tmp_path / 'a.jpg'))
    Image.fromarray(pic).save(str(tmp_path / 'a.gif'))
    for name, word in (('a.jpg', 'JPEG'), ('a.gif', 'GIF')):
        with pytest.raises(M.MarayError) as e:
            M.image_read(str(tmp_path / name))
        assert e.value.code == -3 and word in str(e.value) and 'convert' in str(e.value)
    bad = {'t.bmp': b'BM' + bytes(60), 'h.bmp': b'BM' + struct.pack('<IHHI', 0, 0, 0, 54) + struct.pack('<IiiHHIIiiII', 40, 1 << 19, 1 << 19, 1, 24, 0, 0, 0, 0, 0, 0),
           't.ppm': b'P6 4000 4000 255 abc', 'n.ppm': b'P6 x', 't.qoi': b'qoif' + struct.pack('>IIBB', 9, 9, 3, 0) + bytes(10),
           't.ff': b'farbfeld' + struct.pack('>II', 1 << 20, 1 << 20), 't.tga': bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 100, 0, 100, 0, 24, 0, 0x85]),
           'empty.bin': b'', 'text.txt': b'hello world'}
    for name, data in bad.items():
        p = str(tmp_path / name)
        open(p, 'wb').write(data)
        with pytest.raises(M.MarayError) as e:
            M.image_read(p)
        assert e.value.code in (-3, -7), (name, str(e.value))
    with pytest.raises(M.MarayError) as e:
        M.image_read(str(tmp_path / 'missing.png'))
    assert e.value.code == -2
