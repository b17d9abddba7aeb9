"""Minimal LAS 1.2-1.4 point reader / writer (numpy only) and the TUM-Facade label merge
(SURVEY.md 8f row 3).

The reference reads scenes with `laspy.read` and uses `las.x / las.y / las.z`,
`las.classification` and, per selected feature, `getattr(las, name)` -- red/green/blue in
practice (sem_seg_training.py:137-156, sem_seg_testing.py:133-141).  laspy is not installable in
an offline image, so this module parses the public header block and the fixed part of point
record formats 0-10 itself.  It is NOT pinned against laspy or against TUM-Facade files (neither is
available here): tests read files packed field by field from the specification's tables (independent of
write_las), and cover write/read round trips and the header arithmetic.
"""
import struct

import numpy as np

# byte offsets inside a point record, per point data record format (LAS 1.4 R15 tables)
_CLASS_OFFSET = {**{f: 15 for f in range(0, 6)}, **{f: 16 for f in range(6, 11)}}
_RGB_OFFSET = {2: 20, 3: 28, 5: 28, 7: 30, 8: 30, 10: 30}
_MIN_RECORD = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}

# reference sem_seg_training.py:55 and :159-169 (18 TUM-Facade classes -> 8)
NEW_CLASS_MAPPING = {1: 0, 2: 1, 3: 2, 6: 3, 13: 4, 11: 5, 7: 6, 8: 7}


class LasData:
    """x, y, z float64 (scaled + offset), classification uint8, optional red/green/blue uint16."""

    def __init__(self, x, y, z, classification, red=None, green=None, blue=None, header=None):
        self.x, self.y, self.z = x, y, z
        self.classification = classification
        self.red, self.green, self.blue = red, green, blue
        self.header = header or {}

    def xyz(self):
        """np.vstack((las.x, las.y, las.z)).transpose() of the reference (sem_seg_training.py:138)."""
        return np.vstack((self.x, self.y, self.z)).transpose()


def read_las(path):
    with open(path, "rb") as fh:
        head = fh.read(375)
        if head[:4] != b"LASF":
            raise ValueError("%s is not a LAS file" % path)
        major, minor = head[24], head[25]
        header_size, offset_to_points = struct.unpack_from("<HI", head, 94)
        fmt = head[104] & 0x3F                                    # top bits flag compression (LAZ)
        if head[104] & 0xC0:
            raise NotImplementedError("compressed (LAZ) point records are not supported")
        record_len, legacy_count = struct.unpack_from("<HI", head, 105)
        sx, sy, sz, ox, oy, oz = struct.unpack_from("<6d", head, 131)
        count = legacy_count
        if (major, minor) >= (1, 4) and header_size >= 375:
            count64 = struct.unpack_from("<Q", head, 247)[0]
            count = count64 or legacy_count
        if fmt not in _MIN_RECORD or record_len < _MIN_RECORD[fmt]:
            raise ValueError("unsupported point format %d / record length %d" % (fmt, record_len))
        fh.seek(offset_to_points)
        raw = np.frombuffer(fh.read(count * record_len), dtype=np.uint8)
    if raw.size != count * record_len:
        raise ValueError("truncated point data: expected %d records" % count)
    rec = raw.reshape(count, record_len)
    ints = np.ascontiguousarray(rec[:, :12]).view("<i4")          # X, Y, Z
    x = ints[:, 0] * sx + ox
    y = ints[:, 1] * sy + oy
    z = ints[:, 2] * sz + oz
    cls = rec[:, _CLASS_OFFSET[fmt]].copy()
    if fmt < 6:
        cls &= 0x1F                                               # legacy formats: 5-bit class + flags
    red = green = blue = None
    if fmt in _RGB_OFFSET:
        rgb = np.ascontiguousarray(rec[:, _RGB_OFFSET[fmt]:_RGB_OFFSET[fmt] + 6]).view("<u2")
        red, green, blue = rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy()
    hdr = {"version": (major, minor), "point_format": fmt, "record_length": record_len, "count": count,
           "scale": (sx, sy, sz), "offset": (ox, oy, oz)}
    return LasData(x, y, z, cls, red, green, blue, hdr)


def write_las(path, xyz, classification, rgb=None, scale=0.001, version=(1, 2), point_format=None, extra_bytes=0):
    """Write points as minimal LAS: by default format 2 (with RGB) or 0 under a LAS 1.2 header; point_format
    6 / 7 / 8 writes the LAS 1.4 layouts (375-byte header, 64-bit point count, 8-bit classification at byte 16,
    RGB at byte 30 for 7 and 8).  extra_bytes appends that many user bytes to every record (filled with a pattern),
    as files with "extra bytes" VLR dimensions carry.  For tests and for exporting synthetic scenes."""
    xyz = np.asarray(xyz, dtype=np.float64)
    n = xyz.shape[0]
    fmt = point_format if point_format is not None else (2 if rgb is not None else 0)
    if fmt not in _MIN_RECORD:
        raise ValueError("point format %r" % (fmt,))
    if rgb is not None and fmt not in _RGB_OFFSET:
        raise ValueError("point format %d has no colour" % fmt)
    if fmt >= 6:
        version = (1, 4)
    record_len = _MIN_RECORD[fmt] + int(extra_bytes)
    offset = xyz.min(axis=0) if n else np.zeros(3)
    ints = np.round((xyz - offset) / scale).astype("<i4")
    rec = np.zeros((n, record_len), dtype=np.uint8)
    rec[:, :12] = ints.view(np.uint8).reshape(n, 12)
    cls = np.asarray(classification, dtype=np.uint8)
    rec[:, _CLASS_OFFSET[fmt]] = cls & 0x1F if fmt < 6 else cls
    if rgb is not None:
        o = _RGB_OFFSET[fmt]
        rec[:, o:o + 6] = np.ascontiguousarray(np.asarray(rgb, dtype="<u2")).view(np.uint8).reshape(n, 6)
    if extra_bytes:
        rec[:, _MIN_RECORD[fmt]:] = (np.arange(n)[:, None] + np.arange(extra_bytes)[None, :]) & 0xFF
    header_size = 375 if version >= (1, 4) else 227
    head = bytearray(header_size)
    head[0:4] = b"LASF"
    head[24], head[25] = version
    struct.pack_into("<HI", head, 94, header_size, header_size)
    head[104] = fmt
    struct.pack_into("<HI", head, 105, record_len, n if (fmt < 6 and n < 2 ** 32) else 0)     # legacy count: 0 for formats >= 6
    struct.pack_into("<6d", head, 131, scale, scale, scale, offset[0], offset[1], offset[2])
    mx, mn = (xyz.max(axis=0), xyz.min(axis=0)) if n else (np.zeros(3), np.zeros(3))
    struct.pack_into("<6d", head, 179, mx[0], mn[0], mx[1], mn[1], mx[2], mn[2])
    if header_size >= 375:
        struct.pack_into("<Q", head, 247, n)
    with open(path, "wb") as fh:
        fh.write(bytes(head))
        fh.write(rec.tobytes())


def merge_labels_to_8(labels):
    """The class8 label merge of the reference (sem_seg_training.py:159-169), vectorised.
    Labels outside the mapping become -1 (the reference's dict.get would yield None there)."""
    lab = np.asarray(labels).astype(np.int64).copy()
    lab[(lab == 5) | (lab == 6)] = 6                               # molding + decoration
    lab[(lab == 1) | (lab == 9) | (lab == 15) | (lab == 10)] = 1   # wall, drainpipe, outer ceiling surface, stairs
    lab[(lab == 12) | (lab == 11)] = 11                            # terrain + ground surface
    lab[(lab == 13) | (lab == 16) | (lab == 17)] = 13              # interior, roof, other
    lab[lab == 14] = 2                                             # blinds -> window
    lut = np.full(256, -1, dtype=np.int64)
    for k, v in NEW_CLASS_MAPPING.items():
        lut[k] = v
    return lut[np.clip(lab, 0, 255)]
