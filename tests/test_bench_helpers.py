"""CPU-side helpers of bench.py: the single-member gzip file its gzip leg reads (written slice by slice the way pigz
writes, with a combined CRC-32) must be what gzip.decompress gives back, as one member."""
import gzip
import os
import random
import zlib

from tests.util import REPO


def load_bench():
    import sys

    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import bench  # (under its own name: the helper's worker processes import it again)

    return bench


def test_crc32_combine_is_zlibs():
    b = load_bench()
    r = random.Random(3)
    for _ in range(200):
        x = r.randbytes(r.choice([0, 1, 7, 1000, 70000]))
        y = r.randbytes(r.choice([0, 1, 9, 5000, 131073]))
        assert b._crc32_combine(zlib.crc32(x), zlib.crc32(y), len(y)) == zlib.crc32(x + y)


def test_gz_helper_writes_one_member(tmp_path):
    b = load_bench()
    r = random.Random(4)
    data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(r.choice(b"ACGT") for _ in range(80)), b"I" * 80) for i in range(40000))
    src, dst = tmp_path / "in.fastq", tmp_path / "out.gz"
    src.write_bytes(data + b"trailing bytes that are not part of the prefix")
    b.gz_compress_file(str(src), str(dst), len(data))
    raw = dst.read_bytes()
    assert gzip.decompress(raw) == data
    # one member: the whole file is consumed by ONE raw inflate behind the ten header bytes
    d = zlib.decompressobj(-15)
    out = d.decompress(raw[10:])
    assert out == data and d.eof and len(d.unused_data) == 8
