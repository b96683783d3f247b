"""Deterministic image files for the ingest tests: the same bytes are fed to the product's decoders
(csrc/host/image_io.hpp) and — by tests/golden/make_fixtures.py, where the reference tree exists — to the reference's
real stb_image (oracle/_ref/stb_dump), whose output hashes are committed in tests/golden/image_stb_hashes.json."""
import struct

import numpy as np

from realtimeraytracer_amd import scenes


def _rng_img(h, w, c, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    a[: h // 2, : w // 2] = a[0, 0]            # some flat areas so LZ77 matches + long Huffman runs occur
    return a


def _tga(a, image_type, top_left, rle=False, palette=None, bits16=False):
    """a: (h, w) uint8 for grey / palette indices, (h, w, 3|4) for true colour (RGB[A]); rows given top first"""
    h, w = a.shape[:2]
    if image_type == 1:
        bpp, cmap = 8, b"".join(bytes([c[2], c[1], c[0]]) for c in palette)
        head = struct.pack("<BBBHHBHHHHBB", 0, 1, 9 if rle else 1, 0, len(palette), 24, 0, 0, w, h, 8, 0x20 if top_left else 0)
        pix = [bytes([v]) for v in a.reshape(-1)]
    elif image_type == 3:
        bpp, cmap = 8, b""
        head = struct.pack("<BBBHHBHHHHBB", 0, 0, 11 if rle else 3, 0, 0, 0, 0, 0, w, h, 8, 0x20 if top_left else 0)
        pix = [bytes([v]) for v in a.reshape(-1)]
    else:
        c = a.shape[2]
        cmap = b""
        if bits16:
            bpp = 16
            pix = [struct.pack("<H", ((int(p[0]) >> 3) << 10) | ((int(p[1]) >> 3) << 5) | (int(p[2]) >> 3)) for p in a.reshape(-1, c)]
        else:
            bpp = 8 * c
            pix = [bytes([p[2], p[1], p[0]]) + (bytes([p[3]]) if c == 4 else b"") for p in a.reshape(-1, c)]
        head = struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, bpp, (0x20 if top_left else 0) | (8 if c == 4 else 0))
    rows = [pix[y * w:(y + 1) * w] for y in range(h)]
    if not top_left:
        rows = rows[::-1]
    flat = [q for r in rows for q in r]
    if not rle:
        body = b"".join(flat)
    else:
        body, i = b"", 0
        while i < len(flat):
            run = 1
            while i + run < len(flat) and run < 128 and flat[i + run] == flat[i]:
                run += 1
            if run > 1:
                body += bytes([0x80 | (run - 1)]) + flat[i]; i += run
            else:
                n = 1
                while i + n < len(flat) and n < 128 and (i + n + 1 >= len(flat) or flat[i + n] != flat[i + n + 1]):
                    n += 1
                body += bytes([n - 1]) + b"".join(flat[i:i + n]); i += n
    return head + cmap + body


def _bmp(a, bpp, top_down=False, bitfields=False, palette=None):
    h, w = a.shape[:2]
    stride = ((w * bpp + 31) // 32) * 4
    rows = []
    for y in range(h):
        r = a[y]
        if bpp == 8:
            raw = bytes(r.tolist())
        elif bpp == 24:
            raw = r[:, [2, 1, 0]].tobytes()
        elif bitfields:                                    # masks: R = 0x000000ff, G = 0x0000ff00, B = 0x00ff0000, A = 0xff000000
            raw = r[:, [0, 1, 2, 3]].tobytes()
        else:
            raw = r[:, [2, 1, 0, 3]].tobytes()
        rows.append(raw + b"\0" * (stride - len(raw)))
    if not top_down:
        rows = rows[::-1]
    pal = b"".join(bytes([c[2], c[1], c[0], 0]) for c in palette) if palette else b""
    # bitfields = "v4": BITMAPV4HEADER (108 bytes, the four masks are header fields, alpha used);
    #             True: BITMAPINFOHEADER followed by three masks (no alpha mask -> opaque)
    if bitfields == "v4":
        hdr_size, masks = 108, struct.pack("<IIII", 0xff, 0xff00, 0xff0000, 0xff000000) + b"BGRs" + b"\0" * 48
    elif bitfields:
        hdr_size, masks = 40, struct.pack("<III", 0xff, 0xff00, 0xff0000)
    else:
        hdr_size, masks = 40, b""
    off = 14 + 40 + len(masks) + len(pal)
    info = struct.pack("<IiiHHIIiiII", hdr_size, w, -h if top_down else h, 1, bpp, 3 if bitfields else 0, stride * h, 2835, 2835, len(palette) if palette else 0, 0)
    return b"BM" + struct.pack("<IHHI", off + stride * h, 0, 0, off) + info + masks + pal + b"".join(rows)


def image_cases():
    """[(file name, file bytes)] — PNG colour types x compression levels, PNM, HDR, TGA and BMP variants"""
    import os
    import tempfile
    out = []
    with tempfile.TemporaryDirectory() as d:
        def via(path_writer, name, *args, **kw):
            p = os.path.join(d, name)
            path_writer(p, *args, **kw)
            out.append((name, open(p, "rb").read()))
        for c in (1, 2, 3, 4):
            for level in (0, 1, 9):
                a = _rng_img(37, 53, c, 10 * c + level)
                via(scenes.write_png, f"png_c{c}_l{level}.png", a if c > 1 else a[:, :, 0], level=level)
        a = _rng_img(9, 11, 3, 3)
        via(scenes.write_pnm, "rgb.ppm", a)
        via(scenes.write_pnm, "grey.pgm", a[:, :, 0])
        rng = np.random.default_rng(5)
        f = (rng.random((7, 40, 3)) * np.array([0.2, 1.0, 6.0])).astype(np.float32)
        f[2, 5:30] = f[2, 5]
        via(scenes.write_hdr, "rle.hdr", f, rle=True)
        via(scenes.write_hdr, "flat.hdr", f, rle=False)
    a = _rng_img(13, 21, 4, 7)
    a[3:9, 2:19] = a[3, 2]
    pal = [tuple(int(v) for v in c) for c in _rng_img(1, 40, 3, 9)[0]]
    idx = (a[:, :, 0] % 40).astype(np.uint8)
    for top_left in (True, False):
        for rle in (False, True):
            tag = f"{int(top_left)}{int(rle)}"
            out.append((f"rgba_{tag}.tga", _tga(a, 2, top_left, rle)))
            out.append((f"rgb_{tag}.tga", _tga(a[:, :, :3], 2, top_left, rle)))
            out.append((f"grey_{tag}.tga", _tga(a[:, :, 0], 3, top_left, rle)))
            out.append((f"pal_{tag}.tga", _tga(idx, 1, top_left, rle, palette=pal)))
    out.append(("r5g5b5.tga", _tga(a[:, :, :3], 2, True, bits16=True)))
    b = _rng_img(11, 19, 4, 21)
    z = b.copy(); z[:, :, 3] = 0
    pal2 = [tuple(int(v) for v in c) for c in _rng_img(1, 200, 3, 4)[0]]
    idx2 = (b[:, :, 0] % 200).astype(np.uint8)
    for top_down in (False, True):
        t = int(top_down)
        out.append((f"rgb24_{t}.bmp", _bmp(b[:, :, :3], 24, top_down)))
        out.append((f"rgba32_{t}.bmp", _bmp(b, 32, top_down)))
        out.append((f"zeroalpha32_{t}.bmp", _bmp(z, 32, top_down)))
        out.append((f"bitfields32_{t}.bmp", _bmp(b, 32, top_down, bitfields=True)))
        out.append((f"v4_32_{t}.bmp", _bmp(b, 32, top_down, bitfields="v4")))
        out.append((f"pal8_{t}.bmp", _bmp(idx2, 8, top_down, palette=pal2)))
    out += jpeg_cases()
    out += png_feature_cases()
    return out


def _png(w, h, depth, ctype, rows_of, interlace=False, plte=None, trns=None, level=6):
    """rows_of(x0, y0, dx, dy) -> list of rows (lists of sample tuples) of that pass; filter type cycles 0..4"""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    nsamp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, nsamp * depth // 8)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
    raw = bytearray()
    ft = 0
    for (x0, y0, dx, dy) in passes:
        rows = rows_of(x0, y0, dx, dy)
        prev = None
        for r in rows:
            if not r:
                continue
            flat = [v for px in r for v in px]
            if depth == 16:
                line = b"".join(struct.pack(">H", v) for v in flat)
            elif depth == 8:
                line = bytes(flat)
            else:
                bits = "".join(format(v, "0%db" % depth) for v in flat)
                bits += "0" * ((-len(bits)) % 8)
                line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            up = prev if prev is not None else bytes(len(line))
            f = ft % 5; ft += 1
            out = bytearray()
            for i, v in enumerate(line):
                a = line[i - bpp] if i >= bpp else 0
                b = up[i]; c = up[i - bpp] if i >= bpp else 0
                if f == 0: pr = 0
                elif f == 1: pr = a
                elif f == 2: pr = b
                elif f == 3: pr = (a + b) >> 1
                else:
                    pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                out.append((v - pr) & 255)
            raw += bytes([f]) + out
            prev = line
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        body += chunk(b"PLTE", bytes(v for c in plte for v in c))
    if trns is not None:
        body += chunk(b"tRNS", trns)
    comp = zlib.compress(bytes(raw), level)
    half = len(comp) // 2                                  # two IDAT chunks: the stream continues across chunk boundaries
    return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")


def png_feature_cases():
    """Adam7 interlacing, tRNS in its three forms, 16-bit samples, sub-byte depths — all five filter types in every file"""
    import struct
    out = []
    rng = np.random.default_rng(11)

    def rows_from(a):
        h, w = a.shape[:2]
        return lambda x0, y0, dx, dy: [[tuple(int(v) for v in np.atleast_1d(a[y, x])) for x in range(x0, w, dx)] for y in range(y0, h, dy)]
    a8 = rng.integers(0, 256, (11, 13, 4), dtype=np.uint8)
    a8[2:7, 3:9] = a8[2, 3]
    for il in (False, True):
        t = "adam7" if il else "plain"
        out.append((f"png_{t}_rgba8.png", _png(13, 11, 8, 6, rows_from(a8), il)))
        out.append((f"png_{t}_rgb8_key.png", _png(13, 11, 8, 2, rows_from(a8[:, :, :3]), il, trns=struct.pack(">HHH", *[int(v) for v in a8[2, 3, :3]]))))
        out.append((f"png_{t}_grey8_key.png", _png(13, 11, 8, 0, rows_from(a8[:, :, 0]), il, trns=struct.pack(">H", int(a8[2, 3, 0])))))
        out.append((f"png_{t}_ga8.png", _png(13, 11, 8, 4, rows_from(a8[:, :, :2]), il)))
        a16 = rng.integers(0, 65536, (7, 9, 3)).astype(np.int64); a16[1:4, 2:6] = a16[1, 2]
        out.append((f"png_{t}_rgb16_key.png", _png(9, 7, 16, 2, rows_from(a16), il, trns=struct.pack(">HHH", *[int(v) for v in a16[1, 2]]))))
        out.append((f"png_{t}_rgba16.png", _png(9, 7, 16, 6, rows_from(rng.integers(0, 65536, (7, 9, 4)).astype(np.int64)), il)))
        for depth in (1, 2, 4):
            g = rng.integers(0, 1 << depth, (9, 21)).astype(np.int64)
            out.append((f"png_{t}_grey{depth}.png", _png(21, 9, depth, 0, rows_from(g), il)))
            out.append((f"png_{t}_grey{depth}_key.png", _png(21, 9, depth, 0, rows_from(g), il, trns=struct.pack(">H", 1))))
            pal = [tuple(int(v) for v in c) for c in rng.integers(0, 256, (1 << depth, 3))]
            out.append((f"png_{t}_pal{depth}.png", _png(21, 9, depth, 3, rows_from(g), il, plte=pal)))
        pal = [tuple(int(v) for v in c) for c in rng.integers(0, 256, (200, 3))]
        idx = rng.integers(0, 200, (11, 13)).astype(np.int64)
        out.append((f"png_{t}_pal8_trns.png", _png(13, 11, 8, 3, rows_from(idx), il, plte=pal, trns=bytes(int(v) for v in rng.integers(0, 256, 150)))))
    return out


def _photo(h, w, seed):
    """smooth gradients + a few hard edges + a little noise: runs of zero coefficients, empty blocks and busy blocks"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 9.0 + seed) * np.cos(y / 7.0), 40 + 3.0 * x + 1.5 * y, 255 - 2.5 * y + 20 * np.sin(x / 3.0)], -1)
    img[h // 3: h // 2, w // 4: w // 2] = (250, 20, 60)
    img[: h // 5, : w // 3] = 90                          # a flat area: EOB runs in the progressive scans
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def jpeg_cases():
    import jpeg_writer as J
    a = _photo(45, 67, 1)                                 # neither dimension a multiple of 8 or 16
    b = _photo(32, 48, 2)
    g = _photo(29, 37, 3)[:, :, 0]
    return [
        ("base_grey.jpg", J.encode(g)),
        ("base_444.jpg", J.encode(a, "444")),
        ("base_422.jpg", J.encode(a, "422")),
        ("base_420.jpg", J.encode(a, "420")),
        ("base_440.jpg", J.encode(a, "440")),
        ("base_420_rst3.jpg", J.encode(a, "420", restart=3)),
        ("base_444_scans.jpg", J.encode(b, "444", interleaved=False)),
        ("base_420_scans_rst.jpg", J.encode(a, "420", interleaved=False, restart=5)),
        ("base_rgb_ids.jpg", J.encode(b, "444", adobe_rgb=True)),
        ("base_444_q5.jpg", J.encode(a, "444", scale=400)),       # coarse quantisation: long zero runs
        ("base_444_q1.jpg", J.encode(b, "444", scale=1)),         # quantiser 1: large coefficients, ZRL symbols
        ("prog_grey.jpg", J.encode(g, progressive=True)),
        ("prog_444.jpg", J.encode(b, "444", progressive=True)),
        ("prog_420.jpg", J.encode(a, "420", progressive=True)),
        ("prog_420_rst.jpg", J.encode(a, "420", progressive=True, restart=2)),
        ("prog_422_q1.jpg", J.encode(b, "422", progressive=True, scale=2)),
    ]
