"""SummaryWriter -- the scalar side of `torch.utils.tensorboard.SummaryWriter` the reference logs through
(models/IRNcrop_model.py:78 `SummaryWriter('runs/RHI3')`, :399-400 `add_scalar('PSNR Forward', ...)`), written without the
tensorboard package (absent from the image): `events.out.tfevents.<time>.<host>.<pid>.<n>` files in the TFRecord framing TensorBoard
reads -- per record  u64 length | masked crc32c(length) | payload | masked crc32c(payload) -- whose payloads are `Event` protobufs:

    Event   { 1: wall_time (double)  2: step (int64)  3: file_version (string, first record "brain.Event:2")  5: summary }
    Summary { 1: repeated Value { 1: tag (string)  2: simple_value (float) } }

`read_events(path)` is the matching reader (used by the tests against a file written by the reference's own TensorBoard run)."""
import os
import socket
import struct
import time

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _bytes_field(num, data):
    return _field(num, 2, _varint(len(data)) + data)


def encode_event(wall_time, step=None, file_version=None, scalars=()):
    ev = _field(1, 1, struct.pack("<d", wall_time))
    if step is not None:
        ev += _field(2, 0, _varint(int(step)))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summary = b"".join(_bytes_field(1, _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(v)))) for tag, v in scalars)
        ev += _bytes_field(5, summary)
    return ev


def frame(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


class SummaryWriter(object):
    _count = 0

    def __init__(self, log_dir="runs", filename_suffix=""):
        os.makedirs(log_dir, exist_ok=True)
        SummaryWriter._count += 1
        name = "events.out.tfevents.%010d.%s.%d.%d%s" % (int(time.time()), socket.gethostname(), os.getpid(), SummaryWriter._count, filename_suffix)
        self.path = os.path.join(log_dir, name)
        self._f = open(self.path, "wb")
        self._f.write(frame(encode_event(time.time(), file_version="brain.Event:2")))
        self._f.flush()

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None):
        v = scalar_value.item() if hasattr(scalar_value, "item") else scalar_value
        self._f.write(frame(encode_event(time.time() if walltime is None else walltime, step=global_step, scalars=[(tag, v)])))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


# ---- reader
def _read_varint(buf, i):
    shift, n = 0, 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def _parse(buf):
    """protobuf wire parse -> list of (field number, wire type, value)"""
    out, i = [], 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported wire type %d" % wire)
        out.append((num, wire, v))
    return out


def read_events(path, check_crc=True):
    """-> list of dicts {wall_time, step, file_version, scalars: [(tag, value)]} of a tfevents file (scalar summaries only)"""
    data = open(path, "rb").read()
    events, i = [], 0
    while i + 12 <= len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        (hc,) = struct.unpack("<I", data[i + 8:i + 12])
        payload = data[i + 12:i + 12 + n]
        if len(payload) < n or i + 16 + n > len(data):
            break   # truncated tail
        (pc,) = struct.unpack("<I", data[i + 12 + n:i + 16 + n])
        if check_crc and (hc != masked_crc(head) or pc != masked_crc(payload)):
            raise ValueError("crc mismatch at offset %d" % i)
        i += 16 + n
        ev = {"wall_time": None, "step": 0, "file_version": None, "scalars": []}
        for num, wire, v in _parse(payload):
            if num == 1 and wire == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif num == 2 and wire == 0:
                ev["step"] = v
            elif num == 3 and wire == 2:
                ev["file_version"] = bytes(v).decode()
            elif num == 5 and wire == 2:
                for n2, w2, val in _parse(v):
                    if n2 != 1 or w2 != 2:
                        continue
                    tag, simple = None, None
                    for n3, w3, x in _parse(val):
                        if n3 == 1 and w3 == 2:
                            tag = bytes(x).decode()
                        elif n3 == 2 and w3 == 5:
                            simple = struct.unpack("<f", x)[0]
                    if tag is not None and simple is not None:
                        ev["scalars"].append((tag, simple))
        events.append(ev)
    return events
