"""Texture files: `image::open(file).to_rgb8()` (/root/reference/examples/maray.rs:58-65) takes whatever the `image` crate
knows; maray_image_read takes PNG (every colour type and depth, interlaced too), BMP, PNM, TGA, QOI, farbfeld, GIF and TIFF, told from
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


def _gif(screen, frame, indices, palette, min_size, codes, local=False, gce=None, interlace=False, extra=b''):
    """A GIF by hand: screen (w, h), frame (left, top, w, h), the LZW codes given as (value, bits) pairs."""
    (sw, sh), (fl, ft, fw, fh) = screen, frame
    bits = 0
    nbits = 0
    data = bytearray()
    for v, n in codes:
        bits |= v << nbits
        nbits += n
        while nbits >= 8:
            data.append(bits & 255)
            bits >>= 8
            nbits -= 8
    if nbits:
        data.append(bits & 255)
    blocks = b''.join(bytes([len(data[i:i + 255])]) + bytes(data[i:i + 255]) for i in range(0, len(data), 255)) + b'\0'
    pal = bytes(palette)
    size = {2: 0, 4: 1, 8: 2, 16: 3}[len(palette) // 3]
    out = b'GIF89a' + struct.pack('<HHBBB', sw, sh, 0 if local else 0x80 | size, 0, 0) + (b'' if local else pal)
    if gce is not None:
        out += b'\x21\xF9\x04' + struct.pack('<BHB', 1, 0, gce) + b'\0'
    out += extra + b'\x2C' + struct.pack('<HHHHB', fl, ft, fw, fh, (0x80 | size if local else 0) | (0x40 if interlace else 0)) + (pal if local else b'')
    return out + bytes([min_size]) + blocks + b'\x3B'


def test_gif(tmp_path):
    """GIF: the first frame on the logical screen as `image`'s GifDecoder + to_rgb8 give it -- palette colours (the
    transparent index keeps its colour: only its alpha is 0 and to_rgb8 drops alpha), black outside the frame.  Pillow's
    files (interlaced or not, 2 to 256 colours, long runs and noise: clears and a full code table), an animation (first
    frame), and files made by hand: a frame smaller than the screen, a local colour table, a comment extension, an index
    past the palette."""
    rng = np.random.default_rng(5)
    for (w, h, inter, ncol) in ((37, 23, False, 256), (64, 64, True, 16), (200, 133, False, 2), (1, 1, False, 4), (300, 7, True, 200), (513, 257, False, 256)):
        idx = rng.integers(0, ncol, (h, w), dtype=np.uint8)
        if w > 100:
            idx[:, :w // 2] = idx[0, 0]
        pal = rng.integers(0, 256, (ncol, 3), dtype=np.uint8)
        im = Image.fromarray(idx, 'P')
        im.putpalette(pal.tobytes())
        p = str(tmp_path / ('p_%d_%d.gif' % (w, h)))
        im.save(p, interlace=inter)
        assert np.array_equal(M.image_read(p), pal[idx]), (w, h)
    # transparency: index 3 is transparent and still reads as its palette colour
    idx = rng.integers(0, 8, (20, 30), dtype=np.uint8)
    pal = rng.integers(0, 256, (8, 3), dtype=np.uint8)
    im = Image.fromarray(idx, 'P')
    im.putpalette(pal.tobytes())
    p = str(tmp_path / 'transparent.gif')
    im.save(p, transparency=3)
    assert b'\x21\xF9' in open(p, 'rb').read() and np.array_equal(M.image_read(p), pal[idx])
    # an animation: the first frame
    frames = [Image.fromarray(rng.integers(0, 4, (12, 12), dtype=np.uint8) * 60, 'L').convert('P') for _ in range(3)]
    p = str(tmp_path / 'anim.gif')
    frames[0].save(p, save_all=True, append_images=frames[1:], duration=50)
    assert np.array_equal(M.image_read(p), np.asarray(frames[0].convert('RGB')))
    # by hand: 2-bit codes (min size 2: clear = 4, end = 5), a 2 x 2 frame at (1, 1) of a 4 x 3 screen, local colour table,
    # a comment extension in front, index 3 past a... palette of four: (10,20,30) (40,50,60) (70,80,90) (1,2,3)
    pal4 = [10, 20, 30, 40, 50, 60, 70, 80, 90, 1, 2, 3]
    codes = [(4, 3), (0, 3), (1, 3), (2, 3), (3, 4), (5, 4)]                 # clear, the four pixels, end (the table reaches 8 entries after the third pixel: 4-bit codes from there)
    want = np.zeros((3, 4, 3), np.uint8)
    want[1, 1], want[1, 2], want[2, 1], want[2, 2] = (10, 20, 30), (40, 50, 60), (70, 80, 90), (1, 2, 3)
    for local in (False, True):
        p = str(tmp_path / ('hand_%d.gif' % local))
        open(p, 'wb').write(_gif((4, 3), (1, 1, 2, 2), None, pal4, 2, codes, local=local, gce=2, extra=b'\x21\xFE\x05hello\0'))
        assert np.array_equal(M.image_read(p), want), local
    # ... and a code that is its own definition (KwKwK): pixels 0 0 0 0 0 0 as codes 0, 6 (= "0 0", being defined), 7?  no: 0, 6, 6
    p = str(tmp_path / 'kwkwk.gif')
    open(p, 'wb').write(_gif((5, 1), (0, 0, 5, 1), None, pal4, 2, [(4, 3), (0, 3), (6, 3), (6, 3), (5, 4)]))
    assert np.array_equal(M.image_read(p), np.tile(np.array([10, 20, 30], np.uint8), (1, 5, 1)))
    # broken: a frame outside its screen, a code past the table, data that ends early, no image at all
    bad = {'outside.gif': _gif((4, 3), (3, 1, 2, 2), None, pal4, 2, codes), 'code.gif': _gif((2, 2), (0, 0, 2, 2), None, pal4, 2, [(4, 3), (7, 3), (5, 3)]),
           'early.gif': _gif((2, 2), (0, 0, 2, 2), None, pal4, 2, [(4, 3), (0, 3), (5, 3)]), 'none.gif': b'GIF89a' + struct.pack('<HHBBB', 2, 2, 0, 0, 0) + b'\x3B',
           'trunc.gif': _gif((2, 2), (0, 0, 2, 2), None, pal4, 2, codes)[:-6], 'size.gif': _gif((2, 2), (0, 0, 2, 2), None, pal4, 12, codes)}
    for name, data in bad.items():
        p = str(tmp_path / name)
        open(p, 'wb').write(data)
        with pytest.raises(M.MarayError) as e:
            M.image_read(p)
        assert e.value.code == -3, (name, str(e.value))


def _tiff(big, w, h, fields, strips):
    """A TIFF by hand: header, the strips' bytes, then one IFD of SHORT / LONG fields {tag: value or list}; StripOffsets and
    StripByteCounts are filled in."""
    E = '>' if big else '<'
    body = b''.join(strips)
    offs, at = [], 8
    for st in strips:
        offs.append(at)
        at += len(st)
    fields = dict(fields)
    fields.update({256: w, 257: h, 273: offs, 279: [len(st) for st in strips]})
    ifd_at = 8 + len(body) + (len(body) & 1)
    entries, extra = [], b''
    extra_at = ifd_at + 2 + 12 * len(fields) + 4
    for tag in sorted(fields):
        v = fields[tag] if isinstance(fields[tag], list) else [fields[tag]]
        typ = 4 if tag in (273, 279) or max(v) > 65535 else 3
        raw = b''.join(struct.pack(E + ('I' if typ == 4 else 'H'), x) for x in v)
        if len(raw) <= 4:
            entries.append(struct.pack(E + 'HHI', tag, typ, len(v)) + raw.ljust(4, b'\0'))
        else:
            entries.append(struct.pack(E + 'HHII', tag, typ, len(v), extra_at + len(extra)))
            extra += raw + (b'\0' if len(raw) & 1 else b'')
    return ((b'MM\0*' if big else b'II*\0') + struct.pack(E + 'I', ifd_at) + body + (b'\0' if len(body) & 1 else b'') +
            struct.pack(E + 'H', len(entries)) + b''.join(entries) + struct.pack(E + 'I', 0) + extra)


def test_tiff(tmp_path, pic):
    """TIFF: grey and RGB strips of 8 or 16 bits, either byte order, uncompressed / LZW / Deflate / PackBits, horizontal
    differencing, extra samples dropped, WhiteIsZero inverted -- as `image`'s TiffDecoder + to_rgb8 give them; tiles,
    palettes and other sample widths are named and refused."""
    im = Image.fromarray(pic)
    for name, kw in (('raw', {}), ('lzw', {'compression': 'tiff_lzw'}), ('deflate', {'compression': 'tiff_adobe_deflate'}),
                     ('packbits', {'compression': 'packbits'}), ('lzw_pred', {'compression': 'tiff_lzw', 'tiffinfo': {317: 2}}),
                     ('strips', {'tiffinfo': {278: 5}})):
        p = str(tmp_path / (name + '.tif'))
        im.save(p, **kw)
        assert np.array_equal(M.image_read(p), pic), name
    p = str(tmp_path / 'rgba.tif')
    im.convert('RGBA').save(p)
    assert np.array_equal(M.image_read(p), pic)
    grey = np.asarray(im.convert('L'))
    p = str(tmp_path / 'grey.tif')
    im.convert('L').save(p, compression='tiff_lzw')
    assert np.array_equal(M.image_read(p), np.repeat(grey[:, :, None], 3, axis=2))
    rng = np.random.default_rng(8)
    g16 = rng.integers(0, 65536, (20, 31), dtype=np.uint16)
    p = str(tmp_path / 'grey16.tif')
    Image.fromarray(g16).save(p)
    assert np.array_equal(M.image_read(p)[:, :, 0], ((g16.astype(np.uint32) * 255 + 32767) // 65535).astype(np.uint8))
    big = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    big[:, :200] = big[0, 0]
    p = str(tmp_path / 'big.tif')
    Image.fromarray(big).save(p, compression='tiff_lzw')                 # the code table fills and is cleared
    assert np.array_equal(M.image_read(p), big)
    # by hand: big-endian RGB of 16 bits with horizontal differencing in two strips; little-endian WhiteIsZero grey
    v = rng.integers(0, 65536, (5, 7, 3), dtype=np.uint16)
    d = v.astype(np.int64).copy()
    d[:, 1:] = (v[:, 1:].astype(np.int64) - v[:, :-1]) & 0xFFFF
    rows = [d[y].astype('>u2').tobytes() for y in range(5)]
    want16 = ((v.astype(np.uint32) * 255 + 32767) // 65535).astype(np.uint8)
    p = str(tmp_path / 'be16.tif')
    open(p, 'wb').write(_tiff(True, 7, 5, {258: [16, 16, 16], 259: 1, 262: 2, 277: 3, 278: 3, 317: 2}, [b''.join(rows[:3]), b''.join(rows[3:])]))
    assert np.array_equal(M.image_read(p), want16)
    w0 = rng.integers(0, 256, (4, 6), dtype=np.uint8)
    p = str(tmp_path / 'white0.tif')
    open(p, 'wb').write(_tiff(False, 6, 4, {258: 8, 259: 1, 262: 0, 277: 1, 278: 4}, [w0.tobytes()]))
    assert np.array_equal(M.image_read(p)[:, :, 1], 255 - w0)
    # refused or broken
    raw = pic[:4, :6].tobytes()
    ok = {258: [8, 8, 8], 259: 1, 262: 2, 277: 3, 278: 4}
    bad = {'tiled.tif': _tiff(False, 6, 4, {**ok, 322: 16, 323: 16}, [raw]), 'palette.tif': _tiff(False, 6, 4, {258: 8, 259: 1, 262: 3, 277: 1, 278: 4}, [bytes(24)]),
           'bits4.tif': _tiff(False, 6, 4, {258: 4, 259: 1, 262: 1, 277: 1, 278: 4}, [bytes(12)]), 'short.tif': _tiff(False, 6, 4, ok, [raw[:40]]),
           'comp.tif': _tiff(False, 6, 4, {**ok, 259: 7}, [raw]), 'huge.tif': _tiff(False, 1 << 20, 1 << 20, ok, [raw]),
           'lzw.tif': _tiff(False, 6, 4, {**ok, 259: 5}, [bytes([0x80, 0x3F, 0xFF, 0xFF])]), 'trunc.tif': _tiff(False, 6, 4, ok, [raw])[:60]}
    for name, data in bad.items():
        p = str(tmp_path / name)
        open(p, 'wb').write(data)
        with pytest.raises(M.MarayError) as e:
            M.image_read(p)
        assert e.value.code in (-3, -7), (name, str(e.value))


def test_formats_not_restated_and_broken_files_are_errors(tmp_path, pic):
    Image.fromarray(pic).save(str(tmp_path / 'a.jpg'))
    Image.fromarray(pic).save(str(tmp_path / 'a.webp'), lossless=True)
    for name, word in (('a.jpg', 'JPEG'), ('a.webp', 'WebP')):
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
