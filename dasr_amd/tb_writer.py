"""Minimal TensorBoard event-file writer: `add_scalar` / `add_image` of the reference's `tensorboardX.SummaryWriter`
(codes/SRN/train.py:57-59,112-121,168 and codes/DSN/train.py:245-270) without the tensorboardX / tensorboard packages, which this
image (and an air-gapped training box) does not have.

File format (what TensorBoard reads): a sequence of TFRecords, each `len:uint64 | masked_crc32c(len):uint32 | payload | masked_crc32c(payload)`,
payload = a serialised `tensorflow.Event` protobuf.  Only the fields used here are encoded (by hand, wire format):
    Event   { 1: wall_time (double)  2: step (int64)  3: file_version (string)  5: summary (Summary) }
    Summary { 1: repeated Value { 1: tag (string)  2: simple_value (float)  4: image (Image) } }
    Image   { 1: height  2: width  3: colorspace  4: encoded_image_string (PNG) }
"""
import os
import socket
import struct
import time
import zlib


def _crc32c_table():
    poly = 0x82F63B78
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _crc32c_table()


def _crc_bytes(c, data):
    for b in data:
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


_ZERO_OPS = {}   # L -> [4][256] tables: register value after L zero bytes, per byte lane of the start value (the update is linear over GF(2))


def _zero_advance_tables(L):
    import numpy as np
    if L not in _ZERO_OPS:
        tab = np.array(_TAB, dtype=np.uint32)
        c = (np.arange(256, dtype=np.uint32)[None, :] << (8 * np.arange(4, dtype=np.uint32))[:, None]).reshape(-1)
        for _ in range(L):
            c = tab[c & 0xFF] ^ (c >> 8)
        _ZERO_OPS[L] = [[int(v) for v in row] for row in c.reshape(4, 256)]
    return _ZERO_OPS[L]


def crc32c(data):
    """CRC-32C (Castagnoli) of a bytes-like object.  Records of a few KB go through the byte loop; the MB-sized PNG payloads of add_image
    (validation grids, training samples) are cut into 1024 equal chunks whose registers advance TOGETHER, one numpy table look-up per byte position,
    and are then chained with the 'L zero bytes' operator -- the update is linear over GF(2): state(s, D) = state(s, 0^|D|) xor state(0, D).
    (The per-byte Python loop cost seconds per image on rank 0 while the other data-parallel ranks waited at the next collective)"""
    n = len(data)
    K = 1024
    if n < 16 * K:
        return _crc_bytes(0xFFFFFFFF, data) ^ 0xFFFFFFFF
    import numpy as np
    L = n // K
    a = np.frombuffer(bytes(data[:K * L]) if not isinstance(data, bytes) else data[:K * L], dtype=np.uint8).reshape(K, L)
    tab = np.array(_TAB, dtype=np.uint32)
    c = np.zeros(K, dtype=np.uint32)
    c[0] = 0xFFFFFFFF
    cols = np.ascontiguousarray(a.T)   # [L][K]: one row per byte position
    for j in range(L):
        c = tab[(c ^ cols[j]) & 0xFF] ^ (c >> 8)
    Z = _zero_advance_tables(L)
    s = int(c[0])
    for k in range(1, K):
        s = Z[0][s & 0xFF] ^ Z[1][(s >> 8) & 0xFF] ^ Z[2][(s >> 16) & 0xFF] ^ Z[3][s >> 24] ^ int(c[k])
    return _crc_bytes(s, data[K * L:]) ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _bytes(num, b):
    return _field(num, 2, _varint(len(b)) + b)


def png_encode(chw):
    """float CHW image in [0, 1] (torch tensor or numpy array, C = 1 or 3) -> PNG bytes (8 bit, no external encoder)"""
    import numpy as np
    a = np.asarray(chw.detach().cpu().numpy() if hasattr(chw, 'detach') else chw, dtype=np.float32)
    if a.ndim == 2:
        a = a[None]
    c, h, w = a.shape
    img = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8).transpose(1, 2, 0)
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))

    def chunk(kind, data):
        return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data) & 0xFFFFFFFF)

    ihdr = struct.pack('>IIBBBBB', w, h, 8, 2 if c == 3 else 0, 0, 0, 0)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', ihdr) + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''), h, w, c


class SummaryWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self.f = open(self.path, 'wb')
        self._event(_field(1, 1, struct.pack('<d', time.time())) + _bytes(3, b'brain.Event:2'))

    def _event(self, payload):
        header = struct.pack('<Q', len(payload))
        self.f.write(header + struct.pack('<I', masked_crc(header)) + payload + struct.pack('<I', masked_crc(payload)))
        self.f.flush()

    def _summary_event(self, value, step):
        ev = _field(1, 1, struct.pack('<d', time.time())) + _field(2, 0, _varint(int(step) & 0xFFFFFFFFFFFFFFFF)) + _bytes(5, _bytes(1, value))
        self._event(ev)

    def add_scalar(self, tag, value, step):
        self._summary_event(_bytes(1, tag.encode()) + _field(2, 5, struct.pack('<f', float(value))), step)

    def add_image(self, tag, chw, step):
        png, h, w, c = png_encode(chw)
        image = _field(1, 0, _varint(h)) + _field(2, 0, _varint(w)) + _field(3, 0, _varint(c)) + _bytes(4, png)
        self._summary_event(_bytes(1, tag.encode()) + _bytes(4, image), step)

    def close(self):
        self.f.close()


def read_events(path):
    """parse an event file written by SummaryWriter back (tests): [(step, tag, simple_value | ('image', h, w, png bytes))]; CRCs are checked"""
    out = []
    data = open(path, 'rb').read()
    pos = 0

    def fields(buf):
        i = 0
        while i < len(buf):
            key, n = 0, 0
            while True:
                b = buf[i]
                i += 1
                key |= (b & 0x7F) << n
                n += 7
                if not b & 0x80:
                    break
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, n = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    v |= (b & 0x7F) << n
                    n += 7
                    if not b & 0x80:
                        break
            elif wire == 1:
                v = buf[i:i + 8]
                i += 8
            elif wire == 5:
                v = buf[i:i + 4]
                i += 4
            else:
                ln, n = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    ln |= (b & 0x7F) << n
                    n += 7
                    if not b & 0x80:
                        break
                v = buf[i:i + ln]
                i += ln
            yield num, wire, v

    while pos < len(data):
        header = data[pos:pos + 8]
        ln, = struct.unpack('<Q', header)
        assert struct.unpack('<I', data[pos + 8:pos + 12])[0] == masked_crc(header)
        payload = data[pos + 12:pos + 12 + ln]
        assert struct.unpack('<I', data[pos + 12 + ln:pos + 16 + ln])[0] == masked_crc(payload)
        pos += 16 + ln
        step, summary = 0, None
        for num, wire, v in fields(payload):
            if num == 2:
                step = v
            elif num == 5:
                summary = v
        if summary is None:
            continue
        for num, wire, val in fields(summary):
            tag, item = None, None
            for n2, w2, v2 in fields(val):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2:
                    item = struct.unpack('<f', v2)[0]
                elif n2 == 4:
                    im = {a: b for a, _, b in fields(v2)}
                    item = ('image', im[1], im[2], im[4])
            out.append((step, tag, item))
    return out
