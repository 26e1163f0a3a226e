"""TensorBoard event-file writer — stands in for `tensorboardX.SummaryWriter` (ppo:459-463, not installable here) so that the
reference's scalar names (`charts/*`, `stats/*`, `losses/*`, `eval/*`, ppo:384-406,729-749) land in a `runs/{run_name}` directory
that TensorBoard opens as usual.  Format: TFRecord framing (length, masked CRC32C, payload, masked CRC32C) around hand-encoded
`tensorflow.Event` protobufs (wall_time=1, step=2, file_version=3, summary=5; Summary.Value tag=1, simple_value=2, metadata=9,
tensor=8).  Only what the path logs: add_scalar and add_text."""
import os
import socket
import struct
import threading
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _bytes_field(num: int, data: bytes) -> bytes:
    return _field(num, 2, _varint(len(data)) + data)


def _event(step: int, payload: bytes, wall_time=None) -> bytes:
    e = _field(1, 1, struct.pack("<d", time.time() if wall_time is None else wall_time))
    e += _field(2, 0, _varint(step))
    return e + payload


def _record(data: bytes) -> bytes:
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data))


class SummaryWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}")
        self.f = open(self.path, "ab")
        self.lock = threading.Lock()
        self._write(_event(0, _bytes_field(3, b"brain.Event:2")))

    def _write(self, ev: bytes):
        with self.lock:
            self.f.write(_record(ev))

    def add_scalar(self, tag, value, step):
        val = _bytes_field(1, str(tag).encode()) + _field(2, 5, struct.pack("<f", float(value)))
        self._write(_event(int(step), _bytes_field(5, _bytes_field(1, val))))

    def add_text(self, tag, text, step=0):
        meta = _bytes_field(1, _bytes_field(1, b"text"))                       # SummaryMetadata.plugin_data.plugin_name
        shape = _bytes_field(2, _field(1, 0, _varint(1)))                      # TensorShapeProto.dim{size: 1}
        tensor = _field(1, 0, _varint(7)) + _bytes_field(2, shape) + _bytes_field(8, str(text).encode())   # DT_STRING, shape, string_val
        val = _bytes_field(1, (str(tag) + "/text_summary").encode()) + _bytes_field(9, meta) + _bytes_field(8, tensor)
        self._write(_event(int(step), _bytes_field(5, _bytes_field(1, val))))

    def flush(self):
        with self.lock:
            self.f.flush()

    def close(self):
        with self.lock:
            self.f.close()


def read_scalars(path):
    """Parses an event file written above (checks both CRCs); returns [(step, tag, value)].  Test/inspection helper."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[i + 8:i + 12])[0] == _masked(head), "length CRC"
        rec = data[i + 12:i + 12 + n]
        assert struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] == _masked(rec), "payload CRC"
        i += 16 + n
        step, j = 0, 0

        def rd_varint(buf, k):
            v, s = 0, 0
            while True:
                b = buf[k]; k += 1
                v |= (b & 0x7F) << s; s += 7
                if not b & 0x80:
                    return v, k

        def fields(buf):
            k = 0
            while k < len(buf):
                key, k = rd_varint(buf, k)
                num, wire = key >> 3, key & 7
                if wire == 0:
                    v, k = rd_varint(buf, k)
                elif wire == 1:
                    v, k = buf[k:k + 8], k + 8
                elif wire == 5:
                    v, k = buf[k:k + 4], k + 4
                else:
                    ln, k = rd_varint(buf, k)
                    v, k = buf[k:k + ln], k + ln
                yield num, wire, v
        summary = None
        for num, wire, v in fields(rec):
            if num == 2:
                step = v
            elif num == 5:
                summary = v
        if summary is None:
            continue
        for num, wire, v in fields(summary):
            if num != 1:
                continue
            tag, val = None, None
            for n2, w2, v2 in fields(v):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2 and w2 == 5:
                    val = struct.unpack("<f", v2)[0]
            if val is not None:
                out.append((step, tag, val))
    return out
