"""A TensorBoard event-file writer without TensorBoard: the reference logs its losses and training views through
tensorboardX's SummaryWriter (idr_train.py:11,114-115,516-550,881-895), which this image cannot install.  What a
SummaryWriter puts on disk is small and public: a TFRecord stream (length, masked CRC32C of the length, payload, masked
CRC32C of the payload) of `Event` protocol-buffer messages; scalars are `Summary.Value{tag, simple_value}`, images
`Summary.Value{tag, image{height, width, colorspace, encoded_image_string = PNG}}`.  The handful of fields involved are
written with a 40-line protobuf wire encoder below; `read_events` is the matching reader (the test round-trips through it,
checksums included), so the files open in any TensorBoard.

    w = SummaryWriter(logdir); w.add_scalar('sg_psnr', 31.2, it); w.add_image('train/rgb', chw_float01, it); w.close()
"""
import io
import os
import socket
import struct
import time

import numpy as np

# ---- CRC32C (Castagnoli), table-driven; TFRecord masks it: ((crc >> 15) | (crc << 17)) + 0xa282ead8
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format (varint, 64-bit, length-delimited, 32-bit)
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_double(field, v):
    return _key(field, 1) + struct.pack('<d', float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


def _f_bytes(field, b):
    b = b.encode('utf-8') if isinstance(b, str) else bytes(b)
    return _key(field, 2) + _varint(len(b)) + b


def _event(wall_time, step, payload_field=None, payload=b''):
    """Event{wall_time = 1 (double), step = 2 (int64), file_version = 3 (string) | summary = 5 (message)}"""
    msg = _f_double(1, wall_time) + _f_varint(2, step)
    if payload_field is not None:
        msg += _f_bytes(payload_field, payload)
    return msg


def _png(chw):
    """[C, H, W] or [H, W] float in [0, 1] (or uint8) -> PNG bytes"""
    from PIL import Image
    a = np.asarray(chw)
    if a.ndim == 3:
        a = np.moveaxis(a, 0, -1)
        if a.shape[-1] == 1:
            a = a[..., 0]
    if a.dtype != np.uint8:
        a = (np.clip(a.astype(np.float64), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='PNG')
    return buf.getvalue(), a.shape[0], a.shape[1], (1 if a.ndim == 2 else a.shape[2])


class SummaryWriter:
    """add_scalar / add_image / flush / close of tensorboardX's writer (what the reference calls)."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        self._record(_event(time.time(), 0, 3, 'brain.Event:2'))

    def _record(self, data):
        head = struct.pack('<Q', len(data))
        self._f.write(head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data)))

    def _summary(self, value_msg, step, wall_time=None):
        summary = _f_bytes(1, value_msg)                        # Summary{repeated Value value = 1}
        self._record(_event(time.time() if wall_time is None else wall_time, step, 5, summary))

    def add_scalar(self, tag, value, global_step=0, walltime=None):
        if hasattr(value, 'item'):
            value = value.item()
        self._summary(_f_bytes(1, tag) + _f_float(2, value), global_step, walltime)     # Value{tag = 1, simple_value = 2}

    def add_image(self, tag, img, global_step=0, walltime=None):
        if hasattr(img, 'detach'):
            img = img.detach().cpu().numpy()
        png, h, w, c = _png(img)
        image = _f_varint(1, h) + _f_varint(2, w) + _f_varint(3, c) + _f_bytes(4, png)   # Summary.Image
        self._summary(_f_bytes(1, tag) + _f_bytes(4, image), global_step, walltime)       # Value{tag = 1, image = 4}

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


# ---- reader (tests; a quick look at a run without TensorBoard)
def _parse(buf):
    """flat decode of one message: [(field, wire, value)]"""
    out, i = [], 0
    while i < len(buf):
        k = 0
        shift = 0
        while True:
            b = buf[i]
            i += 1
            k |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        field, wire = k >> 3, k & 7
        if wire == 0:
            v = shift = 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
        elif wire == 1:
            v = struct.unpack('<d', buf[i:i + 8])[0]
            i += 8
        elif wire == 5:
            v = struct.unpack('<f', buf[i:i + 4])[0]
            i += 4
        elif wire == 2:
            n = shift = 0
            while True:
                b = buf[i]
                i += 1
                n |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            v = bytes(buf[i:i + n])
            i += n
        else:
            raise ValueError('wire type %d' % wire)
        out.append((field, wire, v))
    return out


def read_events(path):
    """[{'step', 'wall_time', 'file_version' | 'tag' + ('value' | 'image': (h, w, c, png bytes))}] - checksums verified."""
    data = open(path, 'rb').read()
    events, i = [], 0
    while i < len(data):
        head = data[i:i + 8]
        n = struct.unpack('<Q', head)[0]
        assert struct.unpack('<I', data[i + 8:i + 12])[0] == masked_crc(head), 'length checksum'
        payload = data[i + 12:i + 12 + n]
        assert struct.unpack('<I', data[i + 12 + n:i + 16 + n])[0] == masked_crc(payload), 'payload checksum'
        i += 16 + n
        ev = {'step': 0}
        for field, _, v in _parse(payload):
            if field == 1:
                ev['wall_time'] = v
            elif field == 2:
                ev['step'] = v
            elif field == 3:
                ev['file_version'] = v.decode()
            elif field == 5:
                for f2, _, val in _parse(v):
                    if f2 != 1:
                        continue
                    for f3, _, x in _parse(val):
                        if f3 == 1:
                            ev['tag'] = x.decode()
                        elif f3 == 2:
                            ev['value'] = x
                        elif f3 == 4:
                            img = {a: b for a, _, b in _parse(x)}
                            ev['image'] = (img.get(1), img.get(2), img.get(3), img.get(4))
        events.append(ev)
    return events
