"""PCD and MAT-v5 readers of the C ABI.  Host code: runs without a GPU."""
import struct

import numpy as np
import pytest
import scipy.io

from pcreg_amd import io as pio
from pcreg_amd._lib import PcregError


def test_pcd_ascii_binary_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.normal(0, 30, (1000, 3)).astype(np.float32)
    col = rng.integers(0, 256, (1000, 3)).astype(np.uint8)
    for enc in ("ascii", "binary"):
        p = str(tmp_path / f"c_{enc}.pcd")
        pio.pcwrite(p, xyz, col, encoding=enc)
        got, gc = pio.pcread(p)
        np.testing.assert_array_equal(got, xyz)          # %.9g round-trips float32 exactly
        np.testing.assert_array_equal(gc, col)
        p2 = str(tmp_path / f"n_{enc}.pcd")
        pio.pcwrite(p2, xyz, None, encoding=enc)
        got2, none = pio.pcread(p2)
        assert none is None
        np.testing.assert_array_equal(got2, xyz)


def test_pcd_foreign_headers(tmp_path):
    # a PCL-style file: double coordinates, an extra field before x, float-typed rgb, comment lines
    p = tmp_path / "pcl.pcd"
    rgb = struct.unpack("f", struct.pack("I", (10 << 16) | (20 << 8) | 30))[0]
    p.write_text("# .PCD v.7\nVERSION .7\nFIELDS intensity x y z rgb\nSIZE 4 8 8 8 4\nTYPE F F F F F\nCOUNT 1 1 1 1 1\n"
                 "WIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n"
                 f"0.5 1.25 -2.5 3 {rgb:.9g}\n0.25 1e2 0 -1 {rgb:.9g}\n")
    xyz, col = pio.pcread(str(p))
    np.testing.assert_array_equal(xyz, np.array([[1.25, -2.5, 3], [100, 0, -1]], dtype=np.float32))
    np.testing.assert_array_equal(col, [[10, 20, 30], [10, 20, 30]])
    # binary_compressed: field-major payload behind an LZF stream of literal runs and one back reference
    pts = np.arange(12, dtype=np.float32).reshape(4, 3)
    soa = b"".join(pts[:, k].tobytes() for k in range(3))            # 48 bytes
    lzf = bytes([31]) + soa[:32] + bytes([15]) + soa[32:48]           # two literal runs
    hdr = ("VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary_compressed\n").encode()
    q = tmp_path / "bc.pcd"
    q.write_bytes(hdr + struct.pack("II", len(lzf), 48) + lzf)
    got, _ = pio.pcread(str(q))
    np.testing.assert_array_equal(got, pts)
    # a back reference: 16 zero bytes = literal [0] then copy 15 bytes from distance 1, three times over
    z = np.zeros((4, 3), dtype=np.float32)
    stream = bytes([0, 0]) + bytes([(7 << 5) | 0, 47 - 2 - 7, 0])     # literal 1 byte, then 47 more from dist 1
    q2 = tmp_path / "bz.pcd"
    q2.write_bytes(hdr + struct.pack("II", len(stream), 48) + stream)
    got2, _ = pio.pcread(str(q2))
    np.testing.assert_array_equal(got2, z)


def test_pcd_errors(tmp_path):
    with pytest.raises(PcregError):
        pio.pcread(str(tmp_path / "missing.pcd"))
    bad = tmp_path / "bad.pcd"
    bad.write_text("hello\n")
    with pytest.raises(PcregError):
        pio.pcread(str(bad))
    trunc = tmp_path / "t.pcd"
    trunc.write_text("FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA ascii\n1 2 3\n4 5\n")
    with pytest.raises(PcregError):
        pio.pcread(str(trunc))


@pytest.mark.parametrize("compress", [False, True])
def test_mat_v5_against_scipy(tmp_path, compress):
    rng = np.random.default_rng(1)
    desc = rng.poisson(3.0, (300, 980)).astype(np.float64)
    feat = rng.normal(0, 10, (300, 3))
    small = np.array([[1, 2, 3]], dtype=np.uint8)                     # small-element tags, integer class
    single = rng.normal(0, 1, (7, 5)).astype(np.float32)
    p = str(tmp_path / "d.mat")
    scipy.io.savemat(p, {"featModel": feat, "descModel": desc, "tiny": small, "s": single, "txt": "hello"}, do_compression=compress)
    ref = scipy.io.loadmat(p)
    np.testing.assert_array_equal(pio.load_mat(p, "descModel"), ref["descModel"])
    np.testing.assert_array_equal(pio.load_mat(p, "featModel"), feat)
    np.testing.assert_array_equal(pio.load_mat(p, "tiny"), [[1.0, 2.0, 3.0]])
    np.testing.assert_array_equal(pio.load_mat(p, "s"), single.astype(np.float64))
    np.testing.assert_array_equal(pio.load_mat(p), feat)              # first numeric variable
    with pytest.raises(PcregError):
        pio.load_mat(p, "nope")
    with pytest.raises(PcregError):
        pio.load_mat(p, "txt")                                        # char array: listed, not numeric


def test_mat_rejects_other_containers(tmp_path):
    p = tmp_path / "x.mat"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 200)                 # a v7.3 (HDF5) signature
    with pytest.raises(PcregError):
        pio.load_mat(str(p))
