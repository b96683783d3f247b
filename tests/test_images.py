"""Image decoding of the host layer (csrc/host/image_io.hpp): what the reference gets from stb_image in
core::file::createTextureImage (src/core/file.cppm:272-311) — vertical flip + STBI_grey / STBI_rgb_alpha conversion."""
import os
import struct
import zlib

import numpy as np
import pytest

from realtimeraytracer_amd import host, scenes


def _rng_img(h, w, c, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    a[: h // 2, : w // 2] = a[0, 0]            # some flat areas so LZ77 matches + long Huffman runs occur
    return a


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("level", [0, 1, 9])      # stored blocks, fixed/dynamic Huffman
def test_png_roundtrip_all_colour_types(tmp_path, c, level):
    a = _rng_img(37, 53, c, 10 * c + level)
    p = str(tmp_path / f"t{c}_{level}.png")
    scenes.write_png(p, a if c > 1 else a[:, :, 0], level=level)
    rgba = host.load_image(p, grayscale=False)
    grey = host.load_image(p, grayscale=True)
    src = a[::-1]                                   # stbi_set_flip_vertically_on_load(true)
    if c == 1:
        exp = np.concatenate([src.repeat(3, 2), np.full((37, 53, 1), 255, np.uint8)], 2)
        expg = src[:, :, :1]
    elif c == 2:
        exp = np.concatenate([src[:, :, :1].repeat(3, 2), src[:, :, 1:2]], 2)
        expg = src[:, :, :1]
    else:
        exp = np.concatenate([src[:, :, :3], src[:, :, 3:4] if c == 4 else np.full((37, 53, 1), 255, np.uint8)], 2)
        s = src.astype(np.uint32)
        expg = ((s[:, :, 0] * 77 + s[:, :, 1] * 150 + s[:, :, 2] * 29) >> 8).astype(np.uint8)[:, :, None]
    assert np.array_equal(rgba, exp)
    assert np.array_equal(grey, expg)


def test_png_palette_and_low_bit_depths(tmp_path):
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    w, h = 10, 3
    pal = bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 10, 20, 30])
    idx = np.array([[0, 1, 2, 3, 0, 1, 2, 3, 0, 1], [3, 3, 2, 2, 1, 1, 0, 0, 3, 2], [1] * 10], np.uint8)
    rows = b""
    for y in range(h):                               # 2 bits per pixel, filter 0
        bits = 0
        packed = bytearray()
        for x in range(w):
            bits = (bits << 2) | int(idx[y, x])
            if x % 4 == 3:
                packed.append(bits); bits = 0
        packed.append((bits << 4) & 0xff)            # 10 pixels -> 2.5 bytes: pad the last one
        rows += b"\x00" + bytes(packed)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 2, 3, 0, 0, 0)) + chunk(b"PLTE", pal) + \
        chunk(b"tRNS", bytes([255, 128, 0])) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")
    p = str(tmp_path / "pal.png")
    open(p, "wb").write(png)
    rgba = host.load_image(p)
    palarr = np.frombuffer(pal, np.uint8).reshape(4, 3)
    alpha = np.array([255, 128, 0, 255], np.uint8)
    exp = np.concatenate([palarr[idx], alpha[idx][:, :, None]], 2)[::-1]
    assert np.array_equal(rgba, exp)


def test_pnm_and_hdr(tmp_path):
    a = _rng_img(9, 11, 3, 3)
    scenes.write_pnm(str(tmp_path / "a.ppm"), a)
    scenes.write_pnm(str(tmp_path / "g.pgm"), a[:, :, 0])
    assert np.array_equal(host.load_image(str(tmp_path / "a.ppm"))[:, :, :3], a[::-1])
    assert np.array_equal(host.load_image(str(tmp_path / "g.pgm"), grayscale=True)[:, :, 0], a[::-1, :, 0])
    rng = np.random.default_rng(5)
    f = (rng.random((7, 40, 3)) * np.array([0.2, 1.0, 6.0])).astype(np.float32)
    f[2, 5:30] = f[2, 5]                             # a run for the RLE encoder
    for rle in (True, False):
        p = str(tmp_path / f"s{int(rle)}.hdr")
        rgbe = scenes.write_hdr(p, f, rle=rle)
        got = host.load_image(p)
        # what stb_image's hdr->ldr does to the decoded RGBE: pow(x, 1/2.2) * 255 + 0.5, clamped
        dec = rgbe[..., :3].astype(np.float32) * np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136)[..., None]
        dec[rgbe[..., 3] == 0] = 0
        exp = np.clip(np.power(dec, np.float32(1 / 2.2)) * 255.0 + 0.5, 0, 255).astype(np.uint8)[::-1]
        assert np.abs(got[:, :, :3].astype(int) - exp.astype(int)).max() <= 1     # libm powf vs numpy: at most one code value
        assert np.all(got[:, :, 3] == 255)


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


def test_tga_variants(tmp_path):
    """Truevision TGA: true colour 24/32/16 bit, grey, colour-mapped, RLE, both vertical origins (stb_image: .tga)."""
    a = _rng_img(13, 21, 4, 7)
    a[3:9, 2:19] = a[3, 2]                                # runs for the RLE packets
    flip = lambda x: x[::-1]                              # stbi_set_flip_vertically_on_load(true)
    for top_left in (True, False):
        for rle in (False, True):
            p = str(tmp_path / f"c{int(top_left)}{int(rle)}.tga")
            open(p, "wb").write(_tga(a, 2, top_left, rle))
            assert np.array_equal(host.load_image(p), flip(a)), (top_left, rle)
            open(p, "wb").write(_tga(a[:, :, :3], 2, top_left, rle))
            got = host.load_image(p)
            assert np.array_equal(got[:, :, :3], flip(a[:, :, :3])) and np.all(got[:, :, 3] == 255)
            s = flip(a[:, :, :3]).astype(np.uint32)
            assert np.array_equal(host.load_image(p, grayscale=True)[:, :, 0], ((s[..., 0] * 77 + s[..., 1] * 150 + s[..., 2] * 29) >> 8).astype(np.uint8))
            open(p, "wb").write(_tga(a[:, :, 0], 3, top_left, rle))
            assert np.array_equal(host.load_image(p, grayscale=True)[:, :, 0], flip(a[:, :, 0]))
            assert np.array_equal(host.load_image(p)[:, :, :3], flip(a[:, :, :1]).repeat(3, 2))
            pal = [tuple(int(v) for v in c) for c in _rng_img(1, 40, 3, 9)[0]]
            idx = (a[:, :, 0] % 40).astype(np.uint8)
            open(p, "wb").write(_tga(idx, 1, top_left, rle, palette=pal))
            assert np.array_equal(host.load_image(p)[:, :, :3], flip(np.array(pal, np.uint8)[idx]))
    p = str(tmp_path / "r5g5b5.tga")
    open(p, "wb").write(_tga(a[:, :, :3], 2, True, bits16=True))
    exp = ((a[:, :, :3] >> 3).astype(np.uint32) * 255 // 31).astype(np.uint8)
    assert np.array_equal(host.load_image(p)[:, :, :3], flip(exp))
    bad = str(tmp_path / "trunc.tga")
    open(bad, "wb").write(_tga(a, 2, True)[:100])
    with pytest.raises(host.HostError):
        host.load_image(bad)


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
    hdr_size = 56 if bitfields else 40
    masks = struct.pack("<IIII", 0xff, 0xff00, 0xff0000, 0xff000000) if bitfields else b""
    off = 14 + 40 + len(masks) + len(pal)
    info = struct.pack("<IiiHHIIiiII", hdr_size, w, -h if top_down else h, 1, bpp, 3 if bitfields else 0, stride * h, 2835, 2835, len(palette) if palette else 0, 0)
    return b"BM" + struct.pack("<IHHI", off + stride * h, 0, 0, off) + info + masks + pal + b"".join(rows)


def test_bmp_variants(tmp_path):
    """Windows BMP: 24-bit with row padding, 32-bit BI_RGB (all-zero alpha = opaque), 32-bit BI_BITFIELDS, 8-bit palettised,
    bottom-up and top-down."""
    a = _rng_img(11, 19, 4, 21)
    flip = lambda x: x[::-1]
    for top_down in (False, True):
        p = str(tmp_path / f"t{int(top_down)}.bmp")
        open(p, "wb").write(_bmp(a[:, :, :3], 24, top_down))
        got = host.load_image(p)
        assert np.array_equal(got[:, :, :3], flip(a[:, :, :3])) and np.all(got[:, :, 3] == 255)
        open(p, "wb").write(_bmp(a, 32, top_down))
        assert np.array_equal(host.load_image(p), flip(a))
        z = a.copy(); z[:, :, 3] = 0
        open(p, "wb").write(_bmp(z, 32, top_down))
        got = host.load_image(p)
        assert np.array_equal(got[:, :, :3], flip(a[:, :, :3])) and np.all(got[:, :, 3] == 255)
        open(p, "wb").write(_bmp(a, 32, top_down, bitfields=True))
        assert np.array_equal(host.load_image(p), flip(a))
        pal = [tuple(int(v) for v in c) for c in _rng_img(1, 200, 3, 4)[0]]
        idx = (a[:, :, 0] % 200).astype(np.uint8)
        open(p, "wb").write(_bmp(idx, 8, top_down, palette=pal))
        assert np.array_equal(host.load_image(p)[:, :, :3], flip(np.array(pal, np.uint8)[idx]))
    bad = str(tmp_path / "rle.bmp")
    data = bytearray(_bmp(a[:, :, 0], 8, palette=[(0, 0, 0)] * 256)); data[30] = 1          # BI_RLE8
    open(bad, "wb").write(bytes(data))
    with pytest.raises(host.HostError):
        host.load_image(bad)


def test_bad_images_raise(tmp_path):
    p = str(tmp_path / "x.jpg")
    open(p, "wb").write(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    with pytest.raises(host.HostError):
        host.load_image(p)
    with pytest.raises(host.HostError):
        host.load_image(str(tmp_path / "missing.png"))
    q = str(tmp_path / "trunc.png")
    scenes.write_png(q, _rng_img(16, 16, 3, 1))
    data = open(q, "rb").read()
    open(q, "wb").write(data[: len(data) // 2])
    with pytest.raises(host.HostError):
        host.load_image(q)


def test_texture_table_of_created_scene(scene_cache):
    """create_scene.cppm:71-141: indices start at 2, path-keyed de-dup, specular/metallic R8, colour/opacity RGBA8."""
    s = scenes.textured_room(64, 40)
    d = s.desc
    assert d.numTextures == 2 + 5 and not d.textures[0].pixels and not d.textures[1].pixels
    infos = s.host.objectInfos()
    floor, back, panel, leaf_near, leaf_far, block = infos
    assert floor.usesColorMap and floor.usesSpecularMap and not floor.usesMetallicMap
    assert d.textures[floor.specularIndex].channels == 1 and d.textures[floor.colorIndex].channels == 4
    assert floor.specularIndex == 2 and floor.colorIndex == 3            # specular is visited before colour (:76-106)
    assert panel.usesMetallicMap and d.textures[panel.metallicIndex].channels == 1
    assert leaf_near.usesOpacityMap and leaf_near.opacityIndex == leaf_near.colorIndex == leaf_far.colorIndex   # same path -> one texture
    assert not block.usesColorMap and abs(block.metallic - 0.5) < 1e-7   # unknown_parameter["metallic"]
    assert d.hdri and d.hdri.contents.width == 128 and d.hdri.contents.height == 64 and d.hdri.contents.channels == 4
