"""LAS reader / writer round trip and the class8 label merge.  Unpinned against
laspy / TUM-Facade files (neither exists in the image): round trip + header arithmetic only."""
import numpy as np

from khairil_tum_facade_semantic_segmentation_amd import las


def test_round_trip(tmp_path):
    rs = np.random.RandomState(0)
    n = 5000
    xyz = rs.uniform(0, 1, size=(n, 3)) * [40.0, 25.0, 12.0] + [690000.0, 5336000.0, 500.0]
    cls = rs.randint(0, 18, size=n)
    rgb = rs.randint(0, 65536, size=(n, 3))
    path = str(tmp_path / "scene.las")
    las.write_las(path, xyz, cls, rgb)
    d = las.read_las(path)
    assert d.header["count"] == n and d.header["point_format"] == 2
    assert np.abs(d.xyz() - xyz).max() <= 0.5e-3 + 1e-9           # 1 mm quantisation
    assert np.array_equal(d.classification, cls.astype(np.uint8))
    assert np.array_equal(d.red, rgb[:, 0]) and np.array_equal(d.green, rgb[:, 1]) and np.array_equal(d.blue, rgb[:, 2])
    las.write_las(path, xyz, cls)                                  # format 0: no colour
    d0 = las.read_las(path)
    assert d0.red is None and np.array_equal(d0.classification, cls.astype(np.uint8))


def test_merge_labels_matches_reference_rules():
    got = las.merge_labels_to_8(np.arange(18))
    # sem_seg_training.py:159-169 applied by hand
    want = [-1, 0, 1, 2, -1, 3, 3, 6, 7, 0, 0, 5, 5, 4, 1, 0, 4, 4]
    assert got.tolist() == want


def test_las14_formats_and_extra_bytes_round_trip(tmp_path):
    """LAS 1.4 layouts (formats 6 / 7 / 8: 8-bit classification at byte 16, colour at byte 30, 64-bit count in the
    375-byte header) and records longer than the format's fixed part (extra bytes), as TUM-Facade exports use."""
    rs = np.random.RandomState(1)
    n = 3000
    xyz = rs.uniform(0, 1, size=(n, 3)) * [30.0, 20.0, 15.0] + [690000.0, 5336000.0, 510.0]
    cls = rs.randint(0, 200, size=n)                       # formats >= 6 carry all 8 classification bits
    rgb = rs.randint(0, 65536, size=(n, 3))
    for fmt, extra in ((6, 0), (7, 0), (8, 0), (7, 11), (6, 4), (3, 6)):
        path = str(tmp_path / ("f%d_%d.las" % (fmt, extra)))
        colour = rgb if fmt in (3, 7, 8) else None
        las.write_las(path, xyz, cls if fmt >= 6 else cls % 18, colour, point_format=fmt, extra_bytes=extra)
        d = las.read_las(path)
        assert d.header["point_format"] == fmt and d.header["count"] == n
        assert d.header["record_length"] == las._MIN_RECORD[fmt] + extra
        assert d.header["version"] == ((1, 4) if fmt >= 6 else (1, 2))
        assert np.abs(d.xyz() - xyz).max() <= 0.5e-3 + 1e-9
        assert np.array_equal(d.classification, (cls if fmt >= 6 else cls % 18).astype(np.uint8))
        if colour is None:
            assert d.red is None
        else:
            assert np.array_equal(d.red, rgb[:, 0]) and np.array_equal(d.green, rgb[:, 1]) and np.array_equal(d.blue, rgb[:, 2])
