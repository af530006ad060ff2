"""TensorBoard event files without the tensorboard package (TensorboardLoggerHook,
configs/yunet_n.py:14-17 -> mmcv TensorboardLoggerHook -> SummaryWriter.add_scalar).

File format: a sequence of records  [uint64 length][uint32 masked crc32c(length)][data]
[uint32 masked crc32c(data)], data = a serialized `Event` protobuf:
    Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary { repeated Value value = 1; }   Value { string tag = 1; float simple_value = 2; }
"""
import os
import socket
import struct
import time

_CRC_TABLE = None


def _table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1     # CRC-32C (Castagnoli), reflected
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    t, c = _table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | 0x80 if v else b)
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def event_bytes(wall_time, step=None, tag=None, value=None, file_version=None):
    ev = _varint((1 << 3) | 1) + struct.pack('<d', wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if tag is not None:
        val = _ld(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack('<f', float(value))
        ev += _ld(5, _ld(1, val))
    return ev


def record(data):
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data))


class EventWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, f'events.out.tfevents.{int(time.time())}.{socket.gethostname()}')
        self.f = open(self.path, 'wb')
        self.f.write(record(event_bytes(time.time(), file_version='brain.Event:2')))

    def add_scalar(self, tag, value, step):
        self.f.write(record(event_bytes(time.time(), step, tag, value)))

    def flush(self):
        self.f.flush()

    def close(self):
        self.f.close()


def read_events(path):
    """-> [(step, tag, value)] ; verifies both CRCs of every record (tests)."""
    out = []
    with open(path, 'rb') as f:
        b = f.read()
    i = 0
    while i < len(b):
        n, = struct.unpack_from('<Q', b, i)
        assert struct.unpack_from('<I', b, i + 8)[0] == masked_crc(b[i:i + 8]), 'length crc'
        data = b[i + 12:i + 12 + n]
        assert struct.unpack_from('<I', b, i + 12 + n)[0] == masked_crc(data), 'data crc'
        i += 16 + n
        step, tag, val, j = 0, None, None, 0
        while j < len(data):
            k = data[j]
            j += 1
            f_, w = k >> 3, k & 7
            if w == 1:
                j += 8
            elif w == 0:
                v = s = 0
                while True:
                    c = data[j]
                    j += 1
                    v |= (c & 0x7F) << s
                    s += 7
                    if c < 0x80:
                        break
                if f_ == 2:
                    step = v
            elif w == 2:
                ln = data[j]
                j += 1
                payload = data[j:j + ln]
                j += ln
                if f_ == 5:                                # Summary -> Value -> tag / simple_value
                    inner = payload[2:]
                    tl = inner[1]
                    tag = inner[2:2 + tl].decode()
                    val, = struct.unpack_from('<f', inner, 2 + tl + 1)
        if tag is not None:
            out.append((step, tag, val))
    return out
