"""TensorBoard event files without the tensorboard / tensorboardX dependency.

The reference logs its training scalars with ``tensorboardX.SummaryWriter.add_scalar`` into ``<run>/tb``
(project_ppo/src/ppo.py:13,66,149,892-939).  Neither package is in this image, so this module writes the same on-disk
format directly: a TFRecord stream (u64 length, masked CRC32C of the length, payload, masked CRC32C of the payload) of
``tensorflow.Event`` protobufs, hand-encoded -- field 1 wall_time (double), 2 step (int64), 3 file_version (string),
5 summary { repeated 1 value { 1 tag (string), 2 simple_value (float) } }.  ``read_scalars`` parses it back (tests)."""
import os
import socket
import struct
import time

import numpy as np

_CRC_TABLE = []
for _n in range(256):
    _c = _n
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1   # CRC-32C (Castagnoli), reflected
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


_CRC_TABLE_NP = np.array(_CRC_TABLE, dtype=np.uint32)


def _masked_crc_many(rows):
    """Masked CRC32C of every row of a [K, L] uint8 array: the byte loop runs once over L for all K records (numpy), which
    makes 2,100 scalars cost 12 ms instead of 47 ms of per-byte Python (the default iteration writes ~400: 2-3 ms)."""
    c = np.full(rows.shape[0], 0xFFFFFFFF, dtype=np.uint32)
    for j in range(rows.shape[1]):
        c = _CRC_TABLE_NP[(c ^ rows[:, j]) & 0xFF] ^ (c >> 8)
    c ^= 0xFFFFFFFF
    return (((c >> 15) | (c << 17)) + np.uint32(0xA282EAD8)).astype(np.uint32)


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time, step=None, file_version=None, scalar=None):
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalar is not None:
        tag, value = scalar
        val = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
        ev += _field_bytes(5, _field_bytes(1, val))
    return ev


class SummaryWriter:
    """``add_scalar(tag, value, step)`` / ``flush()`` / ``close()`` -- the subset the reference uses."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
        self._f = open(self.path, "ab")
        self._write(_event(time.time(), file_version="brain.Event:2"))

    def _write(self, payload):
        hdr = struct.pack("<Q", len(payload))
        self._f.write(hdr + struct.pack("<I", _masked_crc(hdr)) + payload + struct.pack("<I", _masked_crc(payload)))

    def add_scalar(self, tag, value, step):
        self._write(_event(time.time(), step=step, scalar=(tag, value)))

    def add_scalars(self, records):
        """Many ``add_scalar`` calls at once: records = iterable of (tag, value, step).  One TFRecord per scalar, exactly as
        add_scalar writes them (every point keeps its own step), but the CRCs are computed for all records of equal length
        together."""
        now = time.time()
        payloads = [_event(now, step=st, scalar=(tag, val)) for tag, val, st in records]
        if not payloads:
            return
        crcs = [0] * len(payloads)
        by_len = {}
        for k, pl in enumerate(payloads):
            by_len.setdefault(len(pl), []).append(k)
        for n, idx in by_len.items():
            rows = np.frombuffer(b"".join(payloads[k] for k in idx), dtype=np.uint8).reshape(len(idx), n)
            for k, c in zip(idx, _masked_crc_many(rows).tolist()):
                crcs[k] = c
        hdr_crc = {n: _masked_crc(struct.pack("<Q", n)) for n in by_len}
        out = bytearray()
        for pl, c in zip(payloads, crcs):
            out += struct.pack("<QI", len(pl), hdr_crc[len(pl)]) + pl + struct.pack("<I", c)
        self._f.write(bytes(out))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_scalars(path):
    """[(tag, step, value)] of an event file; verifies both CRCs of every record."""
    out = []
    data = open(path, "rb").read()
    pos = 0

    def varint(buf, p):
        n = shift = 0
        while True:
            b = buf[p]
            p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p = 0
        while p < len(buf):
            key, p = varint(buf, p)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, p = varint(buf, p)
            elif wt == 1:
                v, p = buf[p:p + 8], p + 8
            elif wt == 5:
                v, p = buf[p:p + 4], p + 4
            else:
                n, p = varint(buf, p)
                v, p = buf[p:p + n], p + n
            yield num, v

    while pos < len(data):
        hdr = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", hdr)
        assert struct.unpack("<I", data[pos + 8:pos + 12])[0] == _masked_crc(hdr), "length CRC"
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(payload), "payload CRC"
        pos += 16 + n
        step, summ = 0, None
        for num, v in fields(payload):
            if num == 2:
                step = v
            elif num == 5:
                summ = v
        if summ is not None:
            for num, v in fields(summ):
                if num == 1:
                    tag = val = None
                    for k, w in fields(v):
                        if k == 1:
                            tag = w.decode()
                        elif k == 2:
                            (val,) = struct.unpack("<f", w)
                    out.append((tag, step, val))
    return out
