"""CPU restatement of the JPEG decode that `reid/utils/data/preprocessor.py:28` performs through Pillow:
`Image.open(fpath).convert('RGB')` on a baseline JPEG = libjpeg(-turbo)'s default decompression -- Huffman decode (jdhuff.c),
dequantisation + the "islow" integer inverse DCT (jidctint.c, CONST_BITS = 13, PASS1_BITS = 2), "fancy" triangle-filter chroma
upsampling (jdsample.c h2v1 / h2v2_fancy_upsample with the context rows of jdmainct.c) and the fixed-point YCbCr -> RGB
conversion (jdcolor.c, SCALEBITS = 16).  The published algorithms are restated here in numpy / plain Python; the result is
pinned against Pillow itself (the reference's decoder) in tests/test_oracle_golden.py on generated files of many sizes,
qualities, chroma subsamplings and restart intervals.

TEST INFRASTRUCTURE ONLY: the product (ssg_amd/jpeg.py + csrc/jpeg.hip) never imports this module; tests and
__graft_entry__.smoke() use it as the checker.

Supported (what the GPU decoder takes; anything else is left to Pillow by the product as well): baseline / extended
sequential Huffman JPEG (SOF0 / SOF1), 8 bit, 1 component or 3 components (YCbCr) with luma sampling 1x1, 2x1 or 2x2 and
1x1 chroma, optional restart intervals.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], np.int32)


class Unsupported(ValueError):
    """not a file of the supported class (progressive, arithmetic, 12 bit, CMYK, unusual sampling ...)"""


def parse(data):
    """marker segments -> dict(width, height, comps=[(id, h, v, tq)], qt={id: [64] natural order}, dc/ac huffman specs
    {id: (bits[16], vals)}, scan=[(comp index, td, ta)], restart_interval, ecs=bytes of the entropy-coded segment incl. RSTn)"""
    d = bytes(data)
    if d[:2] != b"\xff\xd8":
        raise Unsupported("no SOI")
    pos = 2
    out = dict(qt={}, dc={}, ac={}, restart_interval=0, adobe=None)
    while True:
        while pos < len(d) and d[pos] != 0xFF:
            pos += 1
        while pos < len(d) and d[pos] == 0xFF:
            pos += 1
        if pos >= len(d):
            raise Unsupported("no SOS")
        m = d[pos]; pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        L = (d[pos] << 8) | d[pos + 1]
        seg = d[pos + 2:pos + L]
        pos += L
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                q += 1
                if pq:
                    vals = [(seg[q + 2 * i] << 8) | seg[q + 2 * i + 1] for i in range(64)]; q += 128
                else:
                    vals = list(seg[q:q + 64]); q += 64
                nat = np.zeros(64, np.int32)
                nat[ZIGZAG] = vals
                out["qt"][tq] = nat
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                bits = list(seg[q + 1:q + 17]); n = sum(bits)
                vals = list(seg[q + 17:q + 17 + n]); q += 17 + n
                out["ac" if tc else "dc"][th] = (bits, vals)
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported("precision %d" % seg[0])
            out["height"] = (seg[1] << 8) | seg[2]; out["width"] = (seg[3] << 8) | seg[4]
            nc = seg[5]
            out["comps"] = [(seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15, seg[8 + 3 * i]) for i in range(nc)]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("SOF%d (progressive / lossless / arithmetic)" % (m - 0xC0))
        elif m == 0xDD:
            out["restart_interval"] = (seg[0] << 8) | seg[1]
        elif m == 0xEE and seg[:5] == b"Adobe":
            out["adobe"] = seg[11]
        elif m == 0xDA:
            ns = seg[0]
            ids = [c[0] for c in out["comps"]]
            out["scan"] = [(ids.index(seg[1 + 2 * i]), seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)]
            if ns != len(out["comps"]) or seg[1 + 2 * ns] != 0 or seg[2 + 2 * ns] != 63:
                raise Unsupported("multi-scan / spectral selection")
            end = pos
            while True:                       # the entropy-coded segment ends at the first marker that is not RSTn / stuffing
                end = d.index(b"\xff", end)
                nx = d[end + 1] if end + 1 < len(d) else 0xD9
                if nx == 0x00 or 0xD0 <= nx <= 0xD7:
                    end += 2
                    continue
                break
            out["ecs"] = d[pos:end]
            break
        elif m == 0xD9:
            raise Unsupported("EOI before SOS")
    comps = out["comps"]
    if len(comps) == 3:
        if out["adobe"] not in (None, 1) or [c[1:3] for c in comps[1:]] != [(1, 1), (1, 1)] or comps[0][1:3] not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("colour transform / sampling %r" % (comps,))
    elif len(comps) == 1:
        out["comps"] = [(comps[0][0], 1, 1, comps[0][3])]     # a single component is always decoded unsampled
    else:
        raise Unsupported("%d components" % len(comps))
    return out


def huff_tables(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: (mincode[17], maxcode[18], valptr[17], vals) with maxcode[k] = -1 for unused lengths"""
    codes, code, k = [], 0, 0
    sizes = [l for l in range(1, 17) for _ in range(bits[l - 1])]
    si = sizes[0] if sizes else 0
    p = 0
    huffcode = []
    while p < len(sizes):
        while p < len(sizes) and sizes[p] == si:
            huffcode.append(code); code += 1; p += 1
        code <<= 1; si += 1
    maxcode = [-1] * 18; valoff = [0] * 17
    p = 0
    for l in range(1, 17):
        if bits[l - 1]:
            valoff[l] = p - huffcode[p]
            p += bits[l - 1]
            maxcode[l] = huffcode[p - 1]
    maxcode[17] = 0xFFFFF
    return maxcode, valoff, list(vals)


class _Bits:
    def __init__(self, data):
        self.d, self.pos, self.acc, self.n = data, 0, 0, 0

    def _fill(self):
        while self.n <= 24:
            if self.pos < len(self.d):
                b = self.d[self.pos]; self.pos += 1
                if b == 0xFF:
                    nx = self.d[self.pos] if self.pos < len(self.d) else 0
                    if nx == 0:
                        self.pos += 1
                    else:               # a marker: feed zeros (libjpeg "insufficient data" behaviour)
                        self.pos -= 1; b = 0
                        self.d = self.d[:self.pos]
            else:
                b = 0
            self.acc = (self.acc << 8) | b; self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def decode(self, tbl):
        maxcode, valoff, vals = tbl
        code, l = self.get(1), 1
        while code > maxcode[l]:
            code = (code << 1) | self.get(1); l += 1
            if l > 16:
                return 0
        return vals[code + valoff[l]]


def _extend(r, s):
    return r - (1 << s) + 1 if r < (1 << (s - 1)) else r


def decode_coefficients(hdr):
    """-> per component int16 [blocks_h, blocks_w, 64] (natural order, quantised), MCU-padded"""
    comps, W, H = hdr["comps"], hdr["width"], hdr["height"]
    hmax = max(c[1] for c in comps); vmax = max(c[2] for c in comps)
    mcux = (W + 8 * hmax - 1) // (8 * hmax); mcuy = (H + 8 * vmax - 1) // (8 * vmax)
    coef = [np.zeros((mcuy * c[2], mcux * c[1], 64), np.int16) for c in comps]
    dct = {k: huff_tables(*v) for k, v in hdr["dc"].items()}
    act = {k: huff_tables(*v) for k, v in hdr["ac"].items()}
    # split the entropy-coded data at the restart markers
    ecs, segs, start, i = hdr["ecs"], [], 0, 0
    while i + 1 < len(ecs):
        if ecs[i] == 0xFF and 0xD0 <= ecs[i + 1] <= 0xD7:
            segs.append(ecs[start:i]); start = i + 2; i += 2
        elif ecs[i] == 0xFF:
            i += 2
        else:
            i += 1
    segs.append(ecs[start:])
    ri = hdr["restart_interval"] or mcux * mcuy
    mcu = 0
    for seg in segs:
        br = _Bits(seg)
        pred = [0] * len(comps)
        for _ in range(ri):
            if mcu >= mcux * mcuy:
                break
            my, mx = divmod(mcu, mcux)
            for ci, td, ta in hdr["scan"]:
                _, h, v, _ = comps[ci]
                for by in range(v):
                    for bx in range(h):
                        blk = coef[ci][my * v + by, mx * h + bx]
                        s = br.decode(dct[td])
                        if s:
                            pred[ci] += _extend(br.get(s), s)
                        blk[0] = np.int16(pred[ci])
                        k = 1
                        while k < 64:
                            rs = br.decode(act[ta]); r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                blk[ZIGZAG[k & 63]] = np.int16(_extend(br.get(s), s)); k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
            mcu += 1
    return coef


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _range_limit_idct(x):
    """jdmaster.c prepare_range_limit_table, the post-IDCT half: index (x & 1023)"""
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_islow(coef, qt):
    """jidctint.c jpeg_idct_islow on [..., 64] quantised coefficients -> uint8 [..., 8, 8]"""
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)

    def one_d(v0, v1, v2, v3, v4, v5, v6, v7, sh0):
        z2, z3 = v2, v6
        z1 = (z2 + z3) * F["c"]
        tmp2 = z1 + z3 * (-F["h"]); tmp3 = z1 + z2 * F["d"]
        tmp0 = (v0 + v4) << 13; tmp1 = (v0 - v4) << 13
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = v7, v5, v3, v1
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3
        z5 = (z3 + z4) * F["f"]
        t0 = t0 * F["a"]; t1 = t1 * F["j"]; t2 = t2 * F["l"]; t3 = t3 * F["g"]
        z1 = z1 * (-F["e"]); z2 = z2 * (-F["k"]); z3 = z3 * (-F["i"]) + z5; z4 = z4 * (-F["b"]) + z5
        t0 = t0 + z1 + z3; t1 = t1 + z2 + z4; t2 = t2 + z2 + z3; t3 = t3 + z1 + z4
        return [_descale(tmp10 + t3, sh0), _descale(tmp11 + t2, sh0), _descale(tmp12 + t1, sh0), _descale(tmp13 + t0, sh0),
                _descale(tmp13 - t0, sh0), _descale(tmp12 - t1, sh0), _descale(tmp11 - t2, sh0), _descale(tmp10 - t3, sh0)]
    x = coef.astype(np.int64) * qt.astype(np.int64)          # dequantised, [..., 64]
    x = x.reshape(x.shape[:-1] + (8, 8))                     # [row, col]
    ws = np.stack(one_d(*[x[..., r, :] for r in range(8)], 13 - 2), axis=-2)     # pass 1: columns (over the row index)
    out = np.stack(one_d(*[ws[..., :, c] for c in range(8)], 13 + 2 + 3), axis=-1)   # pass 2: rows
    return _range_limit_idct(out)


def _planes(hdr):
    coef = decode_coefficients(hdr)
    planes = []
    for (cid, h, v, tq), cf in zip(hdr["comps"], coef):
        px = idct_islow(cf, hdr["qt"][tq])                   # [by, bx, 8, 8]
        planes.append(px.transpose(0, 2, 1, 3).reshape(cf.shape[0] * 8, cf.shape[1] * 8))
    return planes


def _upsample_h2v1(p, dw):
    """jdsample.c h2v1_fancy_upsample on rows of the first dw columns -> 2*dw columns"""
    a = p[:, :dw].astype(np.int32)
    out = np.empty((a.shape[0], 2 * dw), np.int32)
    if dw <= 2:                                              # libjpeg uses plain replication for such narrow components
        return np.repeat(a, 2, axis=1).astype(np.uint8)
    left = np.concatenate([a[:, :1], a[:, :-1]], axis=1); right = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    out[:, 0::2] = (a * 3 + left + 1) >> 2
    out[:, 1::2] = (a * 3 + right + 2) >> 2
    out[:, 0] = a[:, 0]; out[:, -1] = a[:, -1]
    return out.astype(np.uint8)


def _upsample_h2v2(p, dw, dh):
    """jdsample.c h2v2_fancy_upsample with jdmainct.c's context rows (the first / last real row is its own neighbour)"""
    a = p[:dh, :dw].astype(np.int32)
    if dw <= 2:
        return np.repeat(np.repeat(a, 2, axis=0), 2, axis=1).astype(np.uint8)
    up = np.concatenate([a[:1], a[:-1]], axis=0); dn = np.concatenate([a[1:], a[-1:]], axis=0)
    out = np.empty((2 * dh, 2 * dw), np.int32)
    for v, nb in ((0, up), (1, dn)):
        cs = a * 3 + nb                                       # column sums [dh, dw]
        last = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1); nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        r = np.empty((dh, 2 * dw), np.int32)
        r[:, 0::2] = (cs * 3 + last + 8) >> 4
        r[:, 1::2] = (cs * 3 + nxt + 7) >> 4
        r[:, 0] = (cs[:, 0] * 4 + 8) >> 4; r[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2] = r
    return out.astype(np.uint8)


def decode(data):
    """JPEG file bytes -> uint8 [H, W, 3] == np.asarray(Image.open(...).convert('RGB'))"""
    hdr = parse(data)
    W, H, comps = hdr["width"], hdr["height"], hdr["comps"]
    planes = _planes(hdr)
    if len(comps) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], axis=-1)
    h, v = comps[0][1], comps[0][2]
    y = planes[0][:H, :W].astype(np.int32)
    dw = (W + h - 1) // h; dh = (H + v - 1) // v
    if (h, v) == (1, 1):
        cb, cr = planes[1][:H, :W], planes[2][:H, :W]
    elif (h, v) == (2, 1):
        cb, cr = (_upsample_h2v1(p[:H], dw)[:, :W] for p in planes[1:])
    else:
        cb, cr = (_upsample_h2v2(p, dw, dh)[:H, :W] for p in planes[1:])
    cb = cb.astype(np.int32) - 128; cr = cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)
