"""JPEG decode on the GPU for the extraction loaders -- the decode half of reid/utils/data/preprocessor.py:22-30
(`Image.open(fpath).convert('RGB')`, done by Pillow / libjpeg-turbo on the CPU in the reference's DataLoader workers).

`decode_batch(files)` takes the raw bytes of a batch of image files and returns one uint8 CUDA tensor [H, W, 3] per file whose
bytes equal `np.asarray(Image.open(...).convert('RGB'))`:

* the host walks the marker segments of every file (quantisation / Huffman tables, frame and scan headers, restart interval,
  the position of the entropy-coded data) -- a few hundred bytes of bookkeeping per file;
* `ssg_jpeg_decode_batch` (csrc/jpeg.hip) does the work: Huffman decode, integer inverse DCT, fancy chroma upsampling,
  YCbCr -> RGB, restated from libjpeg's published algorithms (bit-exact);
* files outside the supported class (progressive, arithmetic coding, 12 bit, CMYK / Adobe RGB, sampling factors other than
  1x1 / 2x1 / 2x2, PNG or anything else that is not a JPEG) are decoded by Pillow on the host exactly like the reference does,
  and uploaded; `stats` counts them.
"""
import struct

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

IMG_WORDS, SEG_WORDS = 32, 5
_ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
stats = {"gpu": 0, "pillow": 0}


class NotBaseline(ValueError):
    """the file is left to Pillow"""


class _Header(object):
    __slots__ = ("width", "height", "comps", "qt", "dc", "ac", "scan", "ri", "ecs_start", "ecs_end")


def scan_header(buf):
    """marker walk of one file -> _Header (raises NotBaseline for anything the GPU decoder does not take, including files whose
    marker segments are short, truncated or inconsistent: those go to Pillow, which raises what the reference would raise)"""
    try:
        return _scan_header(buf)
    except (IndexError, struct.error, ValueError, KeyError) as e:
        if isinstance(e, NotBaseline):
            raise
        raise NotBaseline("malformed marker segment (%s)" % (e,))


def _scan_header(buf):
    mv = memoryview(buf)
    n = len(mv)
    if n < 4 or mv[0] != 0xFF or mv[1] != 0xD8:
        raise NotBaseline("not a JPEG")
    h = _Header()
    h.qt, h.dc, h.ac, h.ri, h.comps, h.scan = {}, {}, {}, 0, None, None
    adobe = None
    jfif = False
    p = 2
    while True:
        while p < n and mv[p] != 0xFF:
            p += 1
        while p < n and mv[p] == 0xFF:
            p += 1
        if p >= n:
            raise NotBaseline("no scan")
        marker = mv[p]; p += 1
        if marker == 0x01 or 0xD0 <= marker <= 0xD8:
            continue
        if marker == 0xD9 or p + 2 > n:
            raise NotBaseline("no scan")
        (length,) = struct.unpack_from(">H", mv, p)
        if length < 2 or p + length > n:
            raise NotBaseline("segment length past the end of the file")
        body = bytes(mv[p + 2:p + length])
        p += length
        if marker == 0xDB:                                   # DQT
            q = 0
            while q < len(body):
                prec, tid = body[q] >> 4, body[q] & 15
                q += 1
                if tid > 3 or prec > 1:                      # jdmarker.c get_dqt: JERR_DQT_INDEX
                    raise NotBaseline("bad quantisation table index")
                if prec:
                    vals = struct.unpack_from(">64H", body, q); q += 128
                else:
                    vals = body[q:q + 64]; q += 64
                nat = np.zeros(64, np.uint16)
                nat[_ZZ] = np.frombuffer(bytes(vals), np.uint8) if not prec else np.asarray(vals, np.uint16)
                h.qt[tid] = nat
        elif marker == 0xC4:                                 # DHT
            q = 0
            while q < len(body):
                cls, tid = body[q] >> 4, body[q] & 15
                counts = body[q + 1:q + 17]
                total = sum(counts)
                symbols = bytes(body[q + 17:q + 17 + total])
                # jdhuff.c jpeg_make_d_derived_tbl / jdmarker.c get_dht raise JERR_BAD_HUFF_TABLE for these: leave them to Pillow
                if len(counts) != 16 or total > 256 or len(symbols) != total or cls > 1 or tid > 3 or (cls == 0 and any(v > 15 for v in symbols)):
                    raise NotBaseline("bad Huffman table")
                code = 0
                for length, cnt in enumerate(counts, 1):     # more codes of a length than the code space holds
                    code += cnt
                    if code > (1 << length):
                        raise NotBaseline("bad Huffman table")
                    code <<= 1
                (h.ac if cls else h.dc)[tid] = (bytes(counts), symbols)
                q += 17 + total
        elif marker in (0xC0, 0xC1):                         # baseline / extended sequential, Huffman
            if body[0] != 8:
                raise NotBaseline("%d-bit samples" % body[0])
            h.height, h.width = struct.unpack_from(">HH", body, 1)
            h.comps = [(body[6 + 3 * i], body[7 + 3 * i] >> 4, body[7 + 3 * i] & 15, body[8 + 3 * i]) for i in range(body[5])]
        elif 0xC2 <= marker <= 0xCF and marker not in (0xC4, 0xC8, 0xCC):
            raise NotBaseline("SOF%d" % (marker - 0xC0))
        elif marker == 0xDD:
            (h.ri,) = struct.unpack_from(">H", body, 0)
        elif marker == 0xEE and body[:5] == b"Adobe" and len(body) >= 12:
            adobe = body[11]
        elif marker == 0xE0 and body[:5] == b"JFIF\0":
            jfif = True
        elif marker == 0xDA:                                 # SOS: must be the single scan of a sequential file
            if h.comps is None:
                raise NotBaseline("scan before frame")
            ns = body[0]
            ids = [c[0] for c in h.comps]
            try:
                h.scan = [(ids.index(body[1 + 2 * i]), body[2 + 2 * i] >> 4, body[2 + 2 * i] & 15) for i in range(ns)]
            except ValueError:
                raise NotBaseline("scan component")
            if ns != len(h.comps) or body[1 + 2 * ns] != 0 or body[2 + 2 * ns] != 63 or [s[0] for s in h.scan] != list(range(ns)):
                raise NotBaseline("not one interleaved full scan")
            h.ecs_start = p
            break
    ncomp = len(h.comps)
    if ncomp == 3:
        # jdapimin.c default_decompress_parms: without a JFIF or Adobe marker, component ids 'R','G','B' mean the data IS RGB (no colour
        # transform); any other id triple than (1, 2, 3) makes libjpeg emit a trace message and assume YCbCr, like here
        if not jfif and adobe is None and [c[0] for c in h.comps] == [82, 71, 66]:
            raise NotBaseline("RGB component ids")
        if adobe not in (None, 1) or h.comps[1][1:3] != (1, 1) or h.comps[2][1:3] != (1, 1) or h.comps[0][1:3] not in ((1, 1), (2, 1), (2, 2)):
            raise NotBaseline("colour transform / sampling factors")
    elif ncomp == 1:
        h.comps = [(h.comps[0][0], 1, 1, h.comps[0][3])]
    else:
        raise NotBaseline("%d components" % ncomp)
    if h.width == 0 or h.height == 0 or any(c[3] not in h.qt for c in h.comps) or any(s[1] not in h.dc or s[2] not in h.ac for s in h.scan):
        raise NotBaseline("missing table")
    # end of the entropy-coded data: the first marker that is neither a stuffed zero nor RSTn
    b = bytes(mv[h.ecs_start:])
    q = 0
    while True:
        q = b.find(b"\xff", q)
        if q < 0 or q + 1 >= len(b):
            q = len(b)
            break
        if b[q + 1] == 0 or 0xD0 <= b[q + 1] <= 0xD7:
            q += 2
            continue
        break
    h.ecs_end = h.ecs_start + q
    return h


def _derived(counts, symbols):
    """jdhuff.c jpeg_make_d_derived_tbl -> (look[256] uint16, maxcode[18] int32, valoff[17] int32, vals[256] uint8)"""
    look = np.zeros(256, np.uint16); maxcode = np.full(18, -1, np.int32); valoff = np.zeros(17, np.int32); vals = np.zeros(256, np.uint8)
    vals[:len(symbols)] = np.frombuffer(symbols, np.uint8)
    code, p = 0, 0
    for length in range(1, 17):
        cnt = counts[length - 1]
        if cnt:
            valoff[length] = p - code
            if length <= 8:
                for i in range(cnt):
                    first = (code + i) << (8 - length)
                    look[first:first + (1 << (8 - length))] = (length << 8) | symbols[p + i]
            code += cnt; p += cnt
            maxcode[length] = code - 1
        code <<= 1
    maxcode[17] = 0xFFFFF
    return look, maxcode, valoff, vals


def _pillow_rgb(data):
    import io
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


class _Parsed(object):
    """host-side description of a batch (everything ssg_jpeg_decode_batch needs), numpy arrays + the files left to Pillow"""
    __slots__ = ("kept", "fallback", "dims", "imgs", "segs", "pool", "look", "maxcode", "valoff", "vals", "qts", "blocks", "max_blocks", "plane_bytes",
                 "out_bytes", "max_pixels")


def parse_batch(files, threads=None):
    """marker walk + table building for a batch of files (host only, no device work) -> _Parsed.
    kept = indices of the files the GPU decodes (in batch order), fallback = indices left to Pillow, dims[k] = (H, W) of kept[k].
    Default: the native threaded parser of the C ABI (csrc/jpeg_host.hip: ssg_jpeg_parse_open / _fill / _close; `threads` or
    SSG_JPEG_THREADS, default 4); SSG_JPEG_PARSER=python runs the Python statement below (same result, one core, ~55 us per file)."""
    import os
    if os.environ.get("SSG_JPEG_PARSER", "native") != "python":
        return _parse_batch_native(files, threads)
    return parse_batch_python(files)


def _parse_batch_native(files, threads=None):
    import ctypes
    import os
    L = _lib.lib()
    n = len(files)
    P = _Parsed()
    P.kept, P.fallback, P.dims = [], [], []
    if n == 0:
        return P
    nt = int(threads or os.environ.get("SSG_JPEG_THREADS", "4"))
    keep = [f if isinstance(f, bytes) else bytes(f) for f in files]           # (the C side reads them in place: keep them alive)
    arr = (ctypes.c_char_p * n)(*keep)
    lens = np.fromiter((len(f) for f in keep), np.int64, n)
    counts = np.zeros(10, np.int64); status = np.zeros(n, np.int32)
    h = ctypes.c_void_p()
    check(L.ssg_jpeg_parse_open(ctypes.cast(arr, ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), n, nt, ctypes.byref(h),
                                counts.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.c_void_p)), "ssg_jpeg_parse_open")
    try:
        nk, nseg, npool, ntab, nqt = (int(c) for c in counts[:5])
        P.fallback = np.nonzero(status)[0].tolist()
        P.kept = np.nonzero(status == 0)[0].tolist()
        if nk == 0:
            return P
        P.imgs = np.empty((nk, IMG_WORDS), np.int64); P.segs = np.empty((nseg, SEG_WORDS), np.int64); P.pool = np.empty(npool, np.uint8)
        P.look = np.empty((ntab, 256), np.uint16); P.maxcode = np.empty((ntab, 18), np.int32); P.valoff = np.empty((ntab, 17), np.int32)
        P.vals = np.empty((ntab, 256), np.uint8); P.qts = np.empty((nqt, 64), np.uint16)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)    # noqa: E731
        check(L.ssg_jpeg_parse_fill(h, vp(P.imgs), vp(P.segs), vp(P.pool), vp(P.look), vp(P.maxcode), vp(P.valoff), vp(P.vals), vp(P.qts)), "ssg_jpeg_parse_fill")
    finally:
        L.ssg_jpeg_parse_close(h)
    P.dims = [(int(hh), int(ww)) for ww, hh in P.imgs[:, :2]]
    P.blocks, P.max_blocks, P.plane_bytes, P.out_bytes, P.max_pixels = (int(c) for c in counts[5:10])
    return P


def parse_batch_python(files):
    """the host bookkeeping stated in Python (the checker of the native parser; SSG_JPEG_PARSER=python makes it the product path)"""
    import re
    P = _Parsed()
    hdrs, P.fallback = [], []
    for i, data in enumerate(files):
        try:
            hdrs.append((i, scan_header(data)))
        except NotBaseline:
            P.fallback.append(i)
    P.kept = [i for i, _ in hdrs]
    P.dims = [(h.height, h.width) for _, h in hdrs]
    if not hdrs:
        return P
    huff, huff_ids, qts, qt_ids = [], {}, [], {}

    def huff_id(spec):
        if spec not in huff_ids:
            huff_ids[spec] = len(huff); huff.append(_derived(spec[0], spec[1]))
        return huff_ids[spec]

    def qt_id(q):
        key = q.tobytes()
        if key not in qt_ids:
            qt_ids[key] = len(qts); qts.append(q)
        return qt_ids[key]
    imgs = np.zeros((len(hdrs), IMG_WORDS), np.int64)
    segs, chunks = [], []
    ecs_off = blocks = plane_off = out_off = 0
    max_blocks = max_pixels = 0
    rst = re.compile(b"\xff[\xd0-\xd7]")
    for k, (i, h) in enumerate(hdrs):
        hs, vs = h.comps[0][1], h.comps[0][2]
        mcux = (h.width + 8 * hs - 1) // (8 * hs); mcuy = (h.height + 8 * vs - 1) // (8 * vs)
        imgs[k, :8] = (h.width, h.height, len(h.comps), hs, vs, mcux, mcuy, out_off)
        for ci, (cid, ch, cv, tq) in enumerate(h.comps):
            bw, bh = mcux * ch, mcuy * cv
            _, td, ta = h.scan[ci]
            imgs[k, 8 + 8 * ci:16 + 8 * ci] = (blocks, bw, bh, plane_off, bw * 8, qt_id(h.qt[tq]), huff_id(h.dc[td]), huff_id(h.ac[ta]))
            blocks += bw * bh; plane_off += bw * bh * 64
            max_blocks = max(max_blocks, bw * bh)
        out_off += h.width * h.height * 3
        max_pixels = max(max_pixels, h.width * h.height)
        ecs = bytes(files[i][h.ecs_start:h.ecs_end])
        nmcu = mcux * mcuy
        ri = h.ri or nmcu
        start, m0 = 0, 0
        for mt in rst.finditer(ecs):
            # (a 0xFF byte of the compressed data is always followed by 0x00, so 0xFF 0xDn can only be a restart marker)
            segs.append((k, ecs_off + start, mt.start() - start, m0, min(ri, nmcu - m0)))
            start = mt.end(); m0 += ri
            if m0 >= nmcu:
                break
        if m0 < nmcu:
            segs.append((k, ecs_off + start, len(ecs) - start, m0, min(ri, nmcu - m0)))
        chunks.append(ecs)
        ecs_off += len(ecs)
    P.pool = np.frombuffer(b"".join(chunks) + bytes(64), np.uint8)
    P.segs = np.asarray(segs, np.int64).reshape(-1, SEG_WORDS)
    P.imgs = imgs
    P.look = np.stack([t[0] for t in huff]); P.maxcode = np.stack([t[1] for t in huff]); P.valoff = np.stack([t[2] for t in huff])
    P.vals = np.stack([t[3] for t in huff]); P.qts = np.stack(qts)
    P.blocks, P.max_blocks, P.plane_bytes, P.out_bytes, P.max_pixels = blocks, max_blocks, plane_off, out_off, max_pixels
    return P


class PendingDecode(object):
    """A batch whose GPU decode has been QUEUED: the pixels stay on the device, the per-image status words travel to pinned host
    memory behind the decode on the same stream, and `result()` waits for that copy only (an event recorded right behind it) --
    not for whatever the caller queued afterwards.  A loader that queues batch k + 1 before it asks for batch k's result
    (`GpuBatchLoader`) therefore never waits on the GPU in the common case: the host parse of the next batch overlaps the decode and
    transform of this one, and the blocking read per batch that a damaged file (rare) needs costs nothing when no file is damaged."""

    def __init__(self, files, device, P, out, rgb=None, status_host=None, event=None):
        self.files, self.device, self.P, self.out, self.rgb, self.status_host, self.event = files, device, P, out, rgb, status_host, event

    def result(self, packed=False):
        """list of uint8 CUDA tensors [H, W, 3] in file order -- or, with packed=True and a batch of equally sized files that were all
        decoded on the GPU, ONE tensor [B, H, W, 3]"""
        P, out, files = self.P, self.out, self.files
        if not P.kept:
            return out
        self.event.synchronize()
        damaged = set(np.nonzero(self.status_host.numpy())[0].tolist())
        if packed and not damaged and not P.fallback and all(dm == P.dims[0] for dm in P.dims):
            stats["gpu"] += len(P.kept)
            return self.rgb.view(len(P.kept), P.dims[0][0], P.dims[0][1], 3)
        for k, i in enumerate(P.kept):
            if k in damaged:
                # short / corrupt entropy-coded data: Pillow decodes (or raises "image file is truncated") exactly like the reference
                out[i] = torch.from_numpy(np.array(_pillow_rgb(files[i]))).to(self.device)
                stats["pillow"] += 1; stats["damaged"] = stats.get("damaged", 0) + 1
                continue
            o = int(P.imgs[k, 7])
            hh, ww = P.dims[k]
            out[i] = self.rgb[o:o + ww * hh * 3].view(hh, ww, 3)
        stats["gpu"] += len(P.kept) - len(damaged)
        return out


def decode_batch_async(files, device=None):
    """host parse + upload + the decode kernels of one batch, queued on the current stream -> PendingDecode (no device -> host wait)"""
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = [None] * len(files)
    P = parse_batch(files)
    for i in P.fallback:
        out[i] = torch.from_numpy(np.array(_pillow_rgb(files[i]))).to(device)
        stats["pillow"] += 1
    if not P.kept:
        return PendingDecode(files, device, P, out)
    d = lambda a: torch.from_numpy(np.array(a)).to(device)
    pool_d, segs_d, imgs_d = d(P.pool), d(P.segs), d(P.imgs)
    look_d, maxcode_d, valoff_d, vals_d, qts_d = d(P.look), d(P.maxcode), d(P.valoff), d(P.vals), d(P.qts)
    coef = torch.empty(P.blocks * 64, dtype=torch.int16, device=device)
    planes = torch.empty(max(P.plane_bytes, 1), dtype=torch.uint8, device=device)
    rgb = torch.empty(P.out_bytes, dtype=torch.uint8, device=device)
    status = torch.empty(len(P.kept), dtype=torch.int32, device=device)
    check(L.ssg_jpeg_decode_batch(ptr(pool_d), ptr(segs_d), int(P.segs.shape[0]), ptr(imgs_d), len(P.kept), ptr(look_d), ptr(maxcode_d), ptr(valoff_d), ptr(vals_d),
                                  ptr(qts_d), ptr(coef), P.blocks, P.max_blocks, ptr(planes), P.max_pixels, ptr(rgb), ptr(status), stream()), "ssg_jpeg_decode_batch")
    status_host = torch.empty(len(P.kept), dtype=torch.int32, pin_memory=True)
    status_host.copy_(status, non_blocking=True)       # behind the decode on this stream; the event marks the copy, not later work
    event = torch.cuda.Event()
    event.record()
    return PendingDecode(files, device, P, out, rgb, status_host, event)


def decode_batch(files, device=None, packed=False):
    """list of file contents (bytes) -> list of uint8 CUDA tensors [H, W, 3] (RGB), one per file, in order.
    packed=True: when every file was decoded on the GPU and all share one size (a Market-1501 batch), ONE tensor [B, H, W, 3] is
    returned instead of the list (no per-file views, no stack).  (Blocking form: queue + wait; loaders use `decode_batch_async`.)"""
    return decode_batch_async(files, device).result(packed)
