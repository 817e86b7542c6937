"""Epoch-wise scalars as TensorBoard event files (train_new_task_step2.py:36,115-117,351-355: the reference
logs total / CE / KLD train loss and both domains' validation loss and mIoU through
``torch.utils.tensorboard.SummaryWriter.add_scalar``).  ``tensorboard`` is not a dependency of this package:
the writer below emits the event-file format itself (TFRecord framing with masked CRC-32C, hand-encoded
``Event`` / ``Summary`` protobuf messages) and is used when torch's SummaryWriter cannot be imported."""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc32c(data: bytes) -> int:
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:            # length-delimited field
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _event(wall: float, step: int, file_version=None, tag=None, value=None) -> bytes:
    msg = b"\x09" + struct.pack("<d", wall) + b"\x10" + _varint(step)
    if file_version is not None:
        msg += _ld(3, file_version.encode())
    if tag is not None:
        val = _ld(1, tag.encode()) + b"\x15" + struct.pack("<f", float(value))     # Summary.Value{tag, simple_value}
        msg += _ld(5, _ld(1, val))                                                  # Event.summary{value}
    return msg


class EventFileWriter:
    """``add_scalar(tag, value, step)`` / ``flush()`` / ``close()`` of SummaryWriter, nothing else."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, "wb")
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, data: bytes):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data)))

    def add_scalar(self, tag, value, step):
        self._record(_event(time.time(), int(step), tag=str(tag), value=float(value)))
        self._f.flush()

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def SummaryWriter(logdir):
    """torch's SummaryWriter when tensorboard is installed, else the native event-file writer."""
    try:
        from torch.utils.tensorboard import SummaryWriter as W
        return W(logdir)
    except Exception:
        return EventFileWriter(logdir)


def open_writer(logdir, rank=0):
    """The reference's ``writer = SummaryWriter(logdir)`` at the top of every ``train()`` -- on rank 0 only
    (one process per GPU here; ``nn.DataParallel`` had one process)."""
    return SummaryWriter(logdir) if rank == 0 else None


def add_scalars(writer, info, epoch):
    """The reference's ``for tag, value in info.items(): writer.add_scalar(tag, value, epoch)``."""
    if writer is not None:
        for tag, value in info.items():
            writer.add_scalar(tag, float(value), epoch)


def close_writer(writer):
    """Flush and close (torch's SummaryWriter buffers: the last epochs are lost at exit without it)."""
    if writer is not None:
        writer.flush()
        writer.close()


def read_scalars(path):
    """[(step, tag, value)] of an event file, CRCs verified (tests; a reader for the format above)."""
    out = []
    data = open(path, "rb").read()
    pos = 0

    def varint(buf, i):
        n = s = 0
        while True:
            b = buf[i]
            i += 1
            n |= (b & 0x7F) << s
            s += 7
            if not b & 0x80:
                return n, i

    def fields(buf):
        i = 0
        while i < len(buf):
            key, i = varint(buf, i)
            f, wt = key >> 3, key & 7
            if wt == 0:
                v, i = varint(buf, i)
            elif wt == 1:
                v, i = buf[i:i + 8], i + 8
            elif wt == 5:
                v, i = buf[i:i + 4], i + 4
            else:
                n, i = varint(buf, i)
                v, i = buf[i:i + n], i + n
            yield f, wt, v

    while pos < len(data):
        head = data[pos:pos + 8]
        n, = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[pos + 8:pos + 12])[0] == _masked(head), "length CRC"
        rec = data[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] == _masked(rec), "data CRC"
        pos += 16 + n
        step = 0
        for f, wt, v in fields(rec):
            if f == 2:
                step = v
            elif f == 5:
                for f2, _, val in fields(v):
                    if f2 == 1:
                        tag, x = None, None
                        for f3, wt3, v3 in fields(val):
                            if f3 == 1:
                                tag = v3.decode()
                            elif f3 == 2 and wt3 == 5:
                                x, = struct.unpack("<f", v3)
                        out.append((step, tag, x))
    return out
