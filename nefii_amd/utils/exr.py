"""OpenEXR scanline reader / writer in numpy - the image stack of the reference (imageio + the freeimage plugin,
code/utils/rend_util.py:2-20, code/scripts/render.py:407-442) is not installable here, and the data either side of the
hot path is EXR: `image/*.exr` ground truth (scene_dataset.py:71-80), the environment maps (code/envmaps/*.exr) and
every buffer the renderer writes.

Reads single-part scanline files with NONE / RLE / ZIPS / ZIP / PIZ compression and HALF / FLOAT / UINT channels (what
Blender, freeimage and OpenCV write); tiled, deep and multi-part files and the lossy codecs are refused loudly.  Writes
NONE / ZIPS / ZIP with FLOAT or HALF channels.  `imread` returns what `imageio.imread` returns for these files:
float32 [H, W, C] with channels in R, G, B, A order ([H, W] for a single-channel file).

File layout (OpenEXR file layout document, "scan line" section): magic, version, attributes, a table of one uint64
offset per chunk, then chunks `int32 y, int32 size, bytes`; a chunk's pixel data is, per scan line, per channel in
alphabetical order, `width` little-endian values.  ZIP/RLE store the bytes split into even / odd halves and
delta-coded; PIZ stores 16-bit words per channel, range-compacted through a bitmap LUT, Haar-wavelet transformed
and Huffman coded."""
import struct
import zlib

import numpy as np

MAGIC = 20000630
NO_COMPRESSION, RLE, ZIPS, ZIP, PIZ = 0, 1, 2, 3, 4
LINES_PER_CHUNK = {NO_COMPRESSION: 1, RLE: 1, ZIPS: 1, ZIP: 16, PIZ: 32}
CODEC_NAMES = {0: 'NONE', 1: 'RLE', 2: 'ZIPS', 3: 'ZIP', 4: 'PIZ', 5: 'PXR24', 6: 'B44', 7: 'B44A', 8: 'DWAA', 9: 'DWAB'}
PIXEL_DTYPES = {0: np.dtype('<u4'), 1: np.dtype('<f2'), 2: np.dtype('<f4')}


class ExrError(ValueError):
    pass


# ---- header -------------------------------------------------------------------------------------------------------
def _cstr(b, p):
    e = b.index(b'\0', p)
    return b[p:e].decode('latin-1'), e + 1


def read_header(b):
    """-> (attributes {name: (type, raw bytes)}, offset of the chunk table, version field)"""
    if len(b) < 8 or struct.unpack_from('<i', b, 0)[0] != MAGIC:
        raise ExrError('not an OpenEXR file')
    version = struct.unpack_from('<I', b, 4)[0]
    if version & 0xff != 2:
        raise ExrError('OpenEXR version %d is not supported' % (version & 0xff))
    if version & 0x200:
        raise ExrError('tiled OpenEXR files are not supported')
    if version & 0x1800:
        raise ExrError('deep / multi-part OpenEXR files are not supported')
    p, attrs = 8, {}
    while b[p] != 0:
        name, p = _cstr(b, p)
        typ, p = _cstr(b, p)
        size = struct.unpack_from('<i', b, p)[0]
        p += 4
        attrs[name] = (typ, b[p:p + size])
        p += size
    return attrs, p + 1, version


def parse_channels(raw):
    """chlist -> [(name, pixel type, x sampling, y sampling)] in file (alphabetical) order"""
    p, out = 0, []
    while raw[p] != 0:
        name, p = _cstr(raw, p)
        ptype, _lin, xs, ys = struct.unpack_from('<iB3xii', raw, p)
        p += 16
        if ptype not in PIXEL_DTYPES:
            raise ExrError('unknown pixel type %d in channel %s' % (ptype, name))
        if xs != 1 or ys != 1:
            raise ExrError('sub-sampled channel %s is not supported' % name)
        out.append((name, ptype, xs, ys))
    return out


# ---- ZIP / RLE byte-stream post-processing -------------------------------------------------------------------------
def _unpredict_deinterleave(t):
    """inverse of: split into even / odd bytes, then delta-code (ImfZip.cpp / ImfRle.cpp reconstruct + interleave)"""
    t = np.frombuffer(t, dtype=np.uint8).astype(np.int64)
    n = t.shape[0]
    if n == 0:
        return b''
    d = t - 128
    d[0] = t[0]
    t = (np.cumsum(d) & 0xff).astype(np.uint8)
    out = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _interleave_predict(raw):
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = (t[1:] - t[:-1] + 128 + 256) & 0xff
    return np.concatenate([t[:1], d]).astype(np.uint8).tobytes()


def _rle_decode(src, expect):
    out = bytearray()
    p, n = 0, len(src)
    while p < n:
        c = src[p] - 256 if src[p] > 127 else src[p]
        p += 1
        if c < 0:
            out += src[p:p - c]
            p += -c
        else:
            out += src[p:p + 1] * (c + 1)
            p += 1
    if len(out) != expect:
        raise ExrError('RLE chunk decodes to %d bytes, expected %d' % (len(out), expect))
    return bytes(out)


# ---- PIZ ------------------------------------------------------------------------------------------------------------
HUF_ENCBITS, HUF_DECBITS = 16, 14
HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1
SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN


class _Bits:
    """MSB-first bit reader over bytes"""

    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = (self.c << 8) | (self.d[self.p] if self.p < len(self.d) else 0)
            self.p += 1
            self.lc += 8
        self.lc -= n
        v = self.c >> self.lc
        self.c &= (1 << self.lc) - 1          # keep the accumulator at a few bits (it would grow with the stream)
        return v & ((1 << n) - 1)


def _huf_unpack_table(data, pos, ni, im, iM):
    """code lengths of symbols im..iM: 6 bits each, 59..62 = run of 2..5 zeros, 63 + 8 bits = run of 6..261 zeros"""
    lengths = np.zeros(HUF_ENCSIZE, dtype=np.int64)
    br = _Bits(data, pos)
    s = im
    while s <= iM:
        if br.p - pos > ni:
            raise ExrError('PIZ: Huffman table overruns its block')
        l = br.get(6)
        if l == LONG_ZEROCODE_RUN:
            s += br.get(8) + SHORTEST_LONG_RUN
        elif l >= SHORT_ZEROCODE_RUN:
            s += l - SHORT_ZEROCODE_RUN + 2
        else:
            lengths[s] = l
            s += 1
    return lengths, br.p


def _huf_canonical_codes(lengths):
    """canonical codes: shorter codes have numerically larger prefixes (ImfHuf.cpp hufCanonicalCodeTable)"""
    n = np.bincount(lengths, minlength=59).astype(np.int64)
    c = 0
    first = np.zeros(59, dtype=np.int64)
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        first[i] = c
        c = nc
    codes = np.zeros_like(lengths)
    nxt = first.copy()
    for s in np.nonzero(lengths)[0]:
        l = lengths[s]
        codes[s] = nxt[l]
        nxt[l] += 1
    # Kraft inequality: an over-subscribed table hands out codes that do not fit their length (OpenEXR's
    # hufBuildDecTable: "if (c >> l) invalidTableEntry") - refuse it here, whichever decoder loop runs afterwards
    for l in range(1, 59):
        if n[l] and (int(nxt[l]) - 1) >> l:
            raise ExrError('PIZ: invalid Huffman table (code lengths over-subscribed)')
    return codes


_HOST = [None, False]      # (ctypes library or None, looked for already)


def _host_lib():
    """nefii_amd/csrc/libnefii_host.so (nefii_amd.build.build_host; built on first use when a C compiler is there)"""
    if not _HOST[1]:
        _HOST[1] = True
        try:
            import ctypes
            from .. import build as _build
            lib = ctypes.CDLL(_build.build_host(verbose=False))
            lib.nefii_exr_huf_decode.restype = ctypes.c_long
            lib.nefii_exr_huf_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
            _HOST[0] = lib
        except Exception:                                   # no compiler / read-only tree: the Python loop below
            _HOST[0] = None
    return _HOST[0]


def _huf_decode(data, pos, n_bits, lengths, codes, rlc, n_out):
    lib = _host_lib()
    if lib is not None:
        n_bytes = (n_bits + 7) // 8
        if pos + n_bytes > len(data):
            raise ExrError('PIZ: Huffman bit stream runs past its block')
        buf = np.frombuffer(data, dtype=np.uint8, count=n_bytes, offset=pos).copy()
        ln, cd = np.ascontiguousarray(lengths, dtype=np.int64), np.ascontiguousarray(codes, dtype=np.int64)
        out = np.empty(n_out, dtype=np.uint16)
        got = lib.nefii_exr_huf_decode(buf.ctypes.data, n_bytes, n_bits, ln.ctypes.data, cd.ctypes.data, ln.shape[0], rlc,
                                       out.ctypes.data, n_out)
        if got != n_out:
            raise ExrError('PIZ: Huffman stream ends after %d of %d symbols' % (max(got, 0), n_out))
        return out
    return _huf_decode_py(data, pos, n_bits, lengths, codes, rlc, n_out)


def _huf_decode_py(data, pos, n_bits, lengths, codes, rlc, n_out):
    """bit stream (MSB first) -> n_out uint16 symbols; symbol `rlc` followed by an 8-bit count repeats the previous
    symbol (ImfHuf.cpp hufDecode / getCode).  Codes up to HUF_DECBITS bits resolve through a prefix table, longer ones
    by search."""
    syms = np.nonzero(lengths)[0]
    sym_t = [0] * (1 << HUF_DECBITS)
    len_t = [0] * (1 << HUF_DECBITS)
    long_codes = {}
    for s in syms:
        l, c = int(lengths[s]), int(codes[s])
        if l <= HUF_DECBITS:
            base, span = c << (HUF_DECBITS - l), 1 << (HUF_DECBITS - l)
            sym_t[base:base + span] = [int(s)] * span
            len_t[base:base + span] = [l] * span
        else:
            long_codes[(l, c)] = int(s)
    max_len = int(lengths.max())
    n_bytes = (n_bits + 7) // 8
    if pos + n_bytes > len(data):
        raise ExrError('PIZ: Huffman bit stream runs past its block')
    raw = np.zeros(n_bytes + 8, dtype=np.uint32)
    raw[:n_bytes] = np.frombuffer(data, dtype=np.uint8, count=n_bytes, offset=pos)
    # w[i] = the 32 bits starting at byte i: a k-bit window (k <= 25) at bit position bp is one shift of w[bp >> 3]
    w = ((raw[0:n_bytes + 1] << 24) | (raw[1:n_bytes + 2] << 16) | (raw[2:n_bytes + 3] << 8) | raw[3:n_bytes + 4]).tolist()
    out = []
    bp, o = 0, 0
    while bp < n_bits and o < n_out:
        pre = (w[bp >> 3] >> (18 - (bp & 7))) & 16383
        l = len_t[pre]
        if l:
            s = sym_t[pre]
        else:
            s = None
            window = int.from_bytes(bytes(data[pos + (bp >> 3):pos + (bp >> 3) + 9]).ljust(9, b'\0'), 'big')
            for l in range(HUF_DECBITS + 1, max_len + 1):
                s = long_codes.get((l, (window >> (72 - (bp & 7) - l)) & ((1 << l) - 1)))
                if s is not None:
                    break
            if s is None:
                raise ExrError('PIZ: invalid Huffman code')
        bp += l
        if s == rlc:
            cnt = (w[bp >> 3] >> (24 - (bp & 7))) & 255
            bp += 8
            if o == 0 or o + cnt > n_out:
                raise ExrError('PIZ: run-length symbol out of range')
            out.extend([out[-1]] * cnt)
            o += cnt
        else:
            out.append(s)
            o += 1
    if o != n_out:
        raise ExrError('PIZ: Huffman stream ends after %d of %d symbols' % (o, n_out))
    return np.array(out, dtype=np.uint16)


def _huf_uncompress(data, n_out):
    if n_out == 0:
        return np.zeros(0, dtype=np.uint16)
    if len(data) < 20:
        raise ExrError('PIZ: truncated Huffman block')
    im, iM, _table_len, n_bits, _ = struct.unpack_from('<5I', data, 0)
    if im >= HUF_ENCSIZE or iM >= HUF_ENCSIZE:
        raise ExrError('PIZ: Huffman symbol range out of bounds')
    lengths, pos = _huf_unpack_table(data, 20, len(data) - 20, im, iM)
    codes = _huf_canonical_codes(lengths)
    return _huf_decode(data, pos, n_bits, lengths, codes, iM, n_out)


def _wdec14(l, h):
    """inverse of the 14-bit lifting step (ImfWav.cpp wdec14), on signed 16-bit semantics"""
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    a = ai.astype(np.int16)
    b = (ai - hs).astype(np.int16)
    return a.view(np.uint16), b.view(np.uint16)


def _wdec16(l, h):
    """inverse of the 16-bit (modulo) lifting step (wdec16): A_OFFSET = 1 << 15, MOD_MASK = 0xffff"""
    m = l.astype(np.int32)
    d = h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xffff
    aa = (d + bb - (1 << 15)) & 0xffff
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(buf, nx, ox, ny, oy, mx):
    """in-place inverse 2-D Haar wavelet over the strided view buf[y*oy + x*ox] (ImfWav.cpp wav2Decode)"""
    w14 = mx < (1 << 14)
    dec = _wdec14 if w14 else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    img = np.lib.stride_tricks.as_strided(buf, shape=(ny, nx), strides=(oy * buf.itemsize, ox * buf.itemsize))
    while p >= 1:
        # whole p2 x p2 blocks, top-left corners on the p2 grid
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing='ij')
            p00, p10, p01, p11 = img[Y, X], img[Y + p, X], img[Y, X + p], img[Y + p, X + p]
            i00, i10 = dec(p00, p10)
            i01, i11 = dec(p01, p11)
            a, b = dec(i00, i01)
            c, d = dec(i10, i11)
            img[Y, X], img[Y, X + p], img[Y + p, X], img[Y + p, X + p] = a, b, c, d
        # a column / row the p2 grid leaves uncovered is paired in one direction only: `if (nx & p)` pairs (y, y+p) on
        # the first uncovered column, `if (ny & p)` pairs (x, x+p) on the first uncovered row
        x_edge = len(xs) * p2 if nx & p else None
        y_edge = len(ys) * p2 if ny & p else None
        if x_edge is not None and len(ys):
            a, b = dec(img[ys, x_edge], img[ys + p, x_edge])
            img[ys, x_edge], img[ys + p, x_edge] = a, b
        if y_edge is not None and len(xs):
            a, b = dec(img[y_edge, xs], img[y_edge, xs + p])
            img[y_edge, xs], img[y_edge, xs + p] = a, b
        p2 = p
        p >>= 1


def _piz_decode(src, channels, width, n_lines):
    """one PIZ chunk -> the chunk's uncompressed scan-line-interleaved bytes (ImfPizCompressor.cpp uncompress)"""
    sizes = [PIXEL_DTYPES[t].itemsize // 2 for _, t, _, _ in channels]      # 16-bit words per pixel
    n_words = sum(sizes) * width * n_lines
    min_nz, max_nz = struct.unpack_from('<HH', src, 0)
    p = 4
    bitmap = np.zeros(8192, dtype=np.uint8)
    if min_nz <= max_nz:
        if max_nz >= 8192:
            raise ExrError('PIZ: bitmap range out of bounds')
        bitmap[min_nz:max_nz + 1] = np.frombuffer(src, dtype=np.uint8, count=max_nz - min_nz + 1, offset=p)
        p += max_nz - min_nz + 1
    present = np.unpackbits(bitmap, bitorder='little').astype(bool)
    present[0] = True                                                        # zero is always representable
    lut = np.zeros(65536, dtype=np.uint16)
    vals = np.nonzero(present)[0]
    lut[:len(vals)] = vals
    max_value = len(vals) - 1
    length = struct.unpack_from('<i', src, p)[0]
    p += 4
    if length < 0 or p + length > len(src):
        raise ExrError('PIZ: Huffman block length out of bounds')
    words = _huf_uncompress(src[p:p + length], n_words)
    start = 0
    planes = []
    for size in sizes:
        n = width * n_lines * size
        plane = words[start:start + n]
        for j in range(size):
            _wav2_decode(plane[j:], width, size, n_lines, width * size, max_value)
        planes.append(plane)
        start += n
    for plane in planes:
        plane[:] = lut[plane]
    rows = []
    for y in range(n_lines):
        for plane, size in zip(planes, sizes):
            rows.append(plane[y * width * size:(y + 1) * width * size])
    return np.concatenate(rows).astype('<u2').tobytes() if rows else b''


# ---- reading ----------------------------------------------------------------------------------------------------------
def read_exr(path_or_bytes):
    """-> ({channel name: [H, W] array in its stored dtype}, attributes)"""
    b = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
    attrs, p, _ = read_header(b)
    for need in ('channels', 'compression', 'dataWindow'):
        if need not in attrs:
            raise ExrError('missing header attribute ' + need)
    channels = parse_channels(attrs['channels'][1])
    comp = attrs['compression'][1][0]
    if comp not in LINES_PER_CHUNK:
        raise ExrError('%s-compressed OpenEXR files are not supported' % CODEC_NAMES.get(comp, str(comp)))
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    if W <= 0 or H <= 0:
        raise ExrError('empty data window')
    lpc = LINES_PER_CHUNK[comp]
    n_chunks = (H + lpc - 1) // lpc
    offsets = struct.unpack_from('<%dQ' % n_chunks, b, p)
    line_bytes = sum(PIXEL_DTYPES[t].itemsize for _, t, _, _ in channels) * W
    planes = {name: np.empty((H, W), dtype=PIXEL_DTYPES[t]) for name, t, _, _ in channels}
    for off in offsets:
        y, size = struct.unpack_from('<ii', b, off)
        if off + 8 + size > len(b):
            raise ExrError('chunk at %d runs past the end of the file' % off)
        row0 = y - y0
        if row0 < 0 or row0 >= H:
            raise ExrError('chunk for scan line %d outside the data window' % y)
        n_lines = min(lpc, H - row0)
        expect = line_bytes * n_lines
        data = b[off + 8:off + 8 + size]
        if size < expect:                      # chunks that do not shrink are stored raw
            if comp in (ZIP, ZIPS):
                data = _unpredict_deinterleave(zlib.decompress(data))
            elif comp == RLE:
                data = _unpredict_deinterleave(_rle_decode(data, expect))
            elif comp == PIZ:
                data = _piz_decode(data, channels, W, n_lines)
        if len(data) != expect:
            raise ExrError('chunk at scan line %d decodes to %d bytes, expected %d' % (y, len(data), expect))
        q = 0
        block = np.frombuffer(data, dtype=np.uint8).reshape(n_lines, line_bytes)
        for name, t, _, _ in channels:
            nb = PIXEL_DTYPES[t].itemsize * W
            planes[name][row0:row0 + n_lines] = np.ascontiguousarray(block[:, q:q + nb]).view(PIXEL_DTYPES[t])
            q += nb
    return planes, attrs


def imread(path):
    """imageio.imread for .exr: float32 [H, W, C], channels ordered R, G, B, A (then any others alphabetically);
    a single-channel file gives [H, W]."""
    planes, _ = read_exr(path)
    names = list(planes.keys())
    order = [n for n in ('R', 'G', 'B', 'A') if n in planes]
    order += [n for n in sorted(names) if n not in order]
    # layer-qualified names (Blender multilayer "View Layer.Combined.R") fall back to their last component
    if not any(n in planes for n in ('R', 'G', 'B')):
        short = {n.rsplit('.', 1)[-1]: n for n in names}
        picked = [short[c] for c in ('R', 'G', 'B', 'A') if c in short]
        if picked:
            order = picked
    img = np.stack([planes[n].astype(np.float32) for n in order], axis=-1)
    return img[..., 0] if img.shape[-1] == 1 else img


# ---- writing ----------------------------------------------------------------------------------------------------------
def _attr(name, typ, payload):
    return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(payload)) + payload


def imwrite(path, img, compression='zip', pixel_type='float'):
    """imageio.imwrite for .exr: img [H, W] (channel Y) or [H, W, 3 | 4] (R, G, B[, A])."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[..., None]
    if img.ndim != 3 or img.shape[2] not in (1, 3, 4):
        raise ExrError('imwrite takes [H, W], [H, W, 3] or [H, W, 4], got %s' % (img.shape,))
    H, W, C = img.shape
    comp = {'none': NO_COMPRESSION, 'zips': ZIPS, 'zip': ZIP}.get(compression)
    if comp is None:
        raise ExrError('imwrite compresses with none / zips / zip, not ' + str(compression))
    ptype = {'half': 1, 'float': 2}.get(pixel_type)
    if ptype is None:
        raise ExrError('imwrite stores half or float channels, not ' + str(pixel_type))
    dt = PIXEL_DTYPES[ptype]
    names = {1: ['Y'], 3: ['R', 'G', 'B'], 4: ['R', 'G', 'B', 'A']}[C]
    stored = sorted(range(C), key=lambda i: names[i])
    chlist = b''.join(names[i].encode() + b'\0' + struct.pack('<iB3xii', ptype, 0, 1, 1) for i in stored) + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)
    head = struct.pack('<iI', MAGIC, 2)
    head += _attr('channels', 'chlist', chlist) + _attr('compression', 'compression', bytes([comp]))
    head += _attr('dataWindow', 'box2i', box) + _attr('displayWindow', 'box2i', box)
    head += _attr('lineOrder', 'lineOrder', b'\0') + _attr('pixelAspectRatio', 'float', struct.pack('<f', 1.0))
    head += _attr('screenWindowCenter', 'v2f', struct.pack('<2f', 0.0, 0.0))
    head += _attr('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b'\0'
    lpc = LINES_PER_CHUNK[comp]
    chunks = []
    data = np.stack([img[..., i].astype(dt) for i in stored], axis=1)       # [H, C, W]: per line, per channel
    for r0 in range(0, H, lpc):
        raw = np.ascontiguousarray(data[r0:r0 + lpc]).tobytes()
        out = raw
        if comp != NO_COMPRESSION:
            z = zlib.compress(_interleave_predict(raw))
            out = z if len(z) < len(raw) else raw
        chunks.append(struct.pack('<ii', r0, len(out)) + out)
    table_at = len(head)
    off = table_at + 8 * len(chunks)
    table = b''
    for c in chunks:
        table += struct.pack('<Q', off)
        off += len(c)
    with open(path, 'wb') as f:
        f.write(head + table + b''.join(chunks))
