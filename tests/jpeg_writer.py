"""A small JPEG ENCODER for the ingest tests (baseline and progressive Huffman, ITU-T T.81), written so that the files
exercise every path of csrc/host/jpeg_decode.hpp: interleaved and per-component scans, 4:4:4 / 4:2:2 / 4:2:0 / grey,
restart intervals, spectral selection, successive approximation with DC and AC refinement scans, EOB runs.
Huffman tables are flat (every used symbol gets an 8-bit code), which keeps the encoder short; quality is irrelevant here —
what matters is that the real stb_image and the product decode the same bytes identically."""
import numpy as np
from scipy.fft import dctn

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
QL = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
QC = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)


class Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1); self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xff:
                    self.out.append(0)
                self.acc = 0; self.n = 0

    def flush(self):
        while self.n:
            self.put(1, 1)
        b = bytes(self.out); self.out = bytearray()
        return b


def _seg(marker, payload):
    return bytes([0xff, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload


def _category(v):
    a = abs(int(v)); n = 0
    while a:
        a >>= 1; n += 1
    return n


def _vbits(v, n):
    v = int(v)
    return v if v >= 0 else v + (1 << n) - 1


class _Scan:
    """collects (symbol-table, symbol | raw bits) events, then emits flat Huffman tables + the coded data"""

    def __init__(self):
        self.ev = []                     # ("dc"|"ac", table, symbol) | ("bits", value, n) | ("rst", k)

    def sym(self, kind, table, s):
        self.ev.append((kind, table, s))

    def bits(self, v, n):
        if n:
            self.ev.append(("bits", v, n))

    def emit(self):
        used = {}
        for e in self.ev:
            if e[0] in ("dc", "ac"):
                used.setdefault((e[0], e[1]), set()).add(e[2])
        dht = b""
        code = {}
        for (kind, t), syms in sorted(used.items()):
            syms = sorted(syms)
            assert len(syms) <= 255
            counts = [0] * 16; counts[7] = len(syms)
            dht += bytes([(0x10 if kind == "ac" else 0) | t]) + bytes(counts) + bytes(syms)
            for i, s in enumerate(syms):
                code[(kind, t, s)] = i
        bw = Bits(); data = b""
        for e in self.ev:
            if e[0] == "bits":
                bw.put(e[1], e[2])
            elif e[0] == "rst":
                data += bw.flush() + bytes([0xff, 0xd0 + (e[1] & 7)])
            else:
                bw.put(code[e], 8)
        data += bw.flush()
        return (_seg(0xc4, dht) if dht else b""), data


def _planes(img, sampling):
    """img: (h, w) or (h, w, 3) uint8 -> list of (plane float, h factor, v factor, quant table), level-shifted"""
    if img.ndim == 2:
        return [(img.astype(np.float64) - 128.0, 1, 1, QL)]
    r, g, b = (img[..., k].astype(np.float64) for k in range(3))
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0
    hs, vs = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "440": (1, 2)}[sampling]

    def sub(p):
        h, w = p.shape
        p = np.pad(p, ((0, (-h) % vs), (0, (-w) % hs)), mode="edge")
        return p.reshape(p.shape[0] // vs, vs, p.shape[1] // hs, hs).mean(axis=(1, 3))
    return [(y - 128.0, hs, vs, QL), (sub(cb) - 128.0, 1, 1, QC), (sub(cr) - 128.0, 1, 1, QC)]


def _quantise(plane, h, v, q, mcux, mcuy, scale):
    """-> int coefficients [blocks_y, blocks_x, 64] in zig-zag order, padded to whole MCUs"""
    H, W = mcuy * v * 8, mcux * h * 8
    p = np.pad(plane, ((0, H - plane.shape[0]), (0, W - plane.shape[1])), mode="edge")
    qt = np.clip((q * scale + 50) // 100, 1, 255).astype(np.int64)
    out = np.zeros((H // 8, W // 8, 64), np.int64)
    for by in range(H // 8):
        for bx in range(W // 8):
            c = dctn(p[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8], norm="ortho").reshape(64)
            out[by, bx] = np.round(c / qt)[ZIGZAG].astype(np.int64)
    return out, qt


def encode(img, sampling="444", progressive=False, restart=0, interleaved=True, scale=60, adobe_rgb=False):
    """img: uint8 (h, w) grey or (h, w, 3) RGB.  Returns the bytes of a JFIF file."""
    img = np.asarray(img)
    if adobe_rgb:                        # three components stored as R,G,B with component ids 'R','G','B'
        comps = [(img[..., k].astype(np.float64) - 128.0, 1, 1, QL) for k in range(3)]
    else:
        comps = _planes(img, sampling)
    h, w = img.shape[:2]
    hmax = max(c[1] for c in comps); vmax = max(c[2] for c in comps)
    mcux = -(-w // (8 * hmax)); mcuy = -(-h // (8 * vmax))
    coefs, qts = [], []
    for plane, ch, cv, q in comps:
        c, qt = _quantise(plane, ch, cv, q, mcux, mcuy, scale)
        coefs.append(c); qts.append(qt)
    n = len(comps)
    ids = [ord(x) for x in "RGB"] if adobe_rgb else list(range(1, n + 1))
    tq = [0] + [1] * (n - 1) if not adobe_rgb else [0, 0, 0]
    out = bytes([0xff, 0xd8])
    if not adobe_rgb:
        out += _seg(0xe0, b"JFIF\0" + bytes([1, 1, 0, 0, 1, 0, 1, 0, 0]))
    dqt = b""
    for t in sorted(set(tq)):
        qt = qts[tq.index(t)]
        dqt += bytes([t]) + bytes(int(qt[ZIGZAG[i]]) for i in range(64))
    out += _seg(0xdb, dqt)
    sof = bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([n])
    for k in range(n):
        sof += bytes([ids[k], (comps[k][1] << 4) | comps[k][2], tq[k]])
    out += _seg(0xc2 if progressive else 0xc0, sof)
    if restart:
        out += _seg(0xdd, restart.to_bytes(2, "big"))

    def sos(which, ss, se, ah, al):
        s = bytes([len(which)])
        for k in which:
            s += bytes([ids[k], ((0 if k == 0 else 1) << 4) | (0 if k == 0 else 1)])
        return _seg(0xda, s + bytes([ss, se, (ah << 4) | al]))

    def blocks_of(which):
        """yield lists of (component, by, bx) per MCU, in scan order"""
        if len(which) == 1:
            k = which[0]
            bw = -(-(-(-w * comps[k][1] // hmax)) // 8); bh = -(-(-(-h * comps[k][2] // vmax)) // 8)
            for by in range(bh):
                for bx in range(bw):
                    yield [(k, by, bx)]
        else:
            for my in range(mcuy):
                for mx in range(mcux):
                    yield [(k, my * comps[k][2] + y, mx * comps[k][1] + x) for k in which for y in range(comps[k][2]) for x in range(comps[k][1])]

    def run_scan(which, ss, se, ah, al):
        sc = _Scan()
        pred = [0] * n
        state = {"eobrun": 0, "pending": []}

        def flush_eob(k):
            if state["eobrun"]:
                nb = state["eobrun"].bit_length() - 1
                sc.sym("ac", 0 if k == 0 else 1, nb << 4)
                sc.bits(state["eobrun"] & ((1 << nb) - 1), nb)
                for b in state["pending"]:
                    sc.bits(b, 1)
                state["eobrun"] = 0; state["pending"] = []
        count = 0; rst = 0
        last_k = which[0]
        for mcu in blocks_of(which):
            if restart and count and count % restart == 0:
                flush_eob(last_k)
                sc.ev.append(("rst", rst)); rst += 1
                pred = [0] * n
            count += 1
            for (k, by, bx) in mcu:
                last_k = k
                c = coefs[k][by, bx]
                t = 0 if k == 0 else 1
                if not progressive:
                    d = int(c[0]) - pred[k]; pred[k] = int(c[0])
                    cat = _category(d); sc.sym("dc", t, cat); sc.bits(_vbits(d, cat), cat)
                    r = 0
                    last = max([i for i in range(1, 64) if c[i] != 0], default=0)
                    for i in range(1, last + 1):
                        if c[i] == 0:
                            r += 1; continue
                        while r > 15:
                            sc.sym("ac", t, 0xf0); r -= 16
                        cat = _category(c[i]); sc.sym("ac", t, (r << 4) | cat); sc.bits(_vbits(c[i], cat), cat); r = 0
                    if last < 63:
                        sc.sym("ac", t, 0)
                elif ss == 0:
                    if ah == 0:
                        v = int(c[0]) >> al
                        d = v - pred[k]; pred[k] = v
                        cat = _category(d); sc.sym("dc", t, cat); sc.bits(_vbits(d, cat), cat)
                    else:
                        sc.bits((int(c[0]) >> al) & 1, 1)
                elif ah == 0:                                  # AC first pass (T.81 G.1.2.2)
                    vals = [(abs(int(c[i])) >> al) * (1 if c[i] >= 0 else -1) for i in range(64)]
                    r = 0
                    for i in range(ss, se + 1):
                        if vals[i] == 0:
                            r += 1; continue
                        flush_eob(k)
                        while r > 15:
                            sc.sym("ac", t, 0xf0); r -= 16
                        cat = _category(vals[i]); sc.sym("ac", t, (r << 4) | cat); sc.bits(_vbits(vals[i], cat), cat); r = 0
                    if r > 0:
                        state["eobrun"] += 1
                        if state["eobrun"] == 0x7fff:
                            flush_eob(k)
                else:                                          # AC refinement (T.81 G.1.2.3)
                    av = [abs(int(c[i])) >> al for i in range(64)]
                    eob = max([i for i in range(ss, se + 1) if av[i] == 1], default=-1)
                    r = 0; br = []
                    for i in range(ss, se + 1):
                        if av[i] == 0:
                            r += 1; continue
                        while r > 15 and i <= eob:
                            flush_eob(k)
                            sc.sym("ac", t, 0xf0); r -= 16
                            for b in br:
                                sc.bits(b, 1)
                            br = []
                        if av[i] > 1:
                            br.append(av[i] & 1); continue
                        flush_eob(k)
                        sc.sym("ac", t, (r << 4) | 1); sc.bits(1 if c[i] >= 0 else 0, 1)
                        for b in br:
                            sc.bits(b, 1)
                        br = []; r = 0
                    if r > 0 or br:
                        state["eobrun"] += 1; state["pending"] += br
                        if state["eobrun"] == 0x7fff:
                            flush_eob(k)
        flush_eob(last_k)
        dht, data = sc.emit()
        return dht + sos(which, ss, se, ah, al) + data

    everything = list(range(n))
    if not progressive:
        if interleaved or n == 1:
            out += run_scan(everything, 0, 63, 0, 0)
        else:
            for k in everything:
                out += run_scan([k], 0, 63, 0, 0)
    else:
        out += run_scan(everything, 0, 0, 0, 1)                 # DC, point transform 1
        for k in everything:
            out += run_scan([k], 1, 5, 0, 2)                    # low AC band, two bits to go
            out += run_scan([k], 6, 63, 0, 1)
        for k in everything:
            out += run_scan([k], 1, 5, 2, 1)                    # refinements
        out += run_scan(everything, 0, 0, 1, 0)
        for k in everything:
            out += run_scan([k], 1, 63, 1, 0)
    return out + bytes([0xff, 0xd9])
