"""The many-core gzip reader of the drop-in programs (fastq_utils_amd/host/fq_pgzip.h) against zlib's gzread.

CPU only.  tests/cxx/pgzip_check.cpp reads a file both ways and compares: the same bytes, or the same refusal (zlib's
text).  The files: every deflate flavour zlib writes (levels, stored / fixed / dynamic blocks, flush points as pigz
makes them), several members, members that end inside a chunk, bytes behind the last member, header fields, cut and
damaged files - with chunk sizes far below the product's, so that a file of a few MB is dozens of chunks that must be
found, joined and stitched.
"""
import gzip
import os
import random
import re
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "tests", "cxx")


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    d = tmp_path_factory.mktemp("pgzip")
    path = str(d / "pgzip_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", path,
                    os.path.join(CXX, "pgzip_check.cpp"), "-lz", "-pthread"], check=True)
    return path


def fastq_text(n, seed, read_len=(50, 150), names="SYN"):
    r = random.Random(seed)
    out = []
    for i in range(n):
        L = r.randint(*read_len)
        seq = "".join(r.choice("ACGTN") if r.random() < 0.02 else r.choice("ACGT") for _ in range(L))
        q = "".join(chr(33 + min(41, max(2, int(r.gauss(34, 5))))) for _ in range(L))
        out.append("@%s.%d lane:%d:%d\n%s\n+\n%s\n" % (names, i, r.randint(1, 8), r.randint(1, 99999), seq, q))
    return "".join(out).encode()


def deflate_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush=zlib.Z_SYNC_FLUSH, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = []
    for o in range(0, len(data), flush_every):
        out.append(c.compress(data[o:o + flush_every]))
        out.append(c.flush(flush))
    out.append(c.flush())
    return b"".join(out)


def member(data, header=None, **kw):
    if header is None:
        header = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"
    return header + deflate_raw(data, **kw) + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def run(check, path, threads, chunk, read_size=None):
    cmd = [check, str(path), str(threads), str(chunk)] + ([str(read_size)] if read_size else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = p.stdout.strip()
    stats = {k: v for k, v in re.findall(r"(\w+)=(\S+)", line)}
    return p.returncode, line, stats


TEXT = {}


def text(n, seed):
    if (n, seed) not in TEXT:
        TEXT[(n, seed)] = fastq_text(n, seed)
    return TEXT[(n, seed)]


@pytest.mark.parametrize("level", [1, 4, 6, 9])
@pytest.mark.parametrize("chunk", [16384, 100000, 1 << 20])
def test_single_member_levels(check, tmp_path, level, chunk):
    data = text(30000, 1)
    f = tmp_path / "a.gz"
    f.write_bytes(gzip.compress(data, level))
    rc, line, st = run(check, f, 4, chunk)
    assert rc == 0, line
    assert int(st["bytes"]) == len(data)
    assert st["fell_back"] == "0", line
    if chunk < 200000:
        assert int(st["joined"]) > int(st["batches"]), line  # chunks beyond each round's first were found and joined


@pytest.mark.parametrize("threads", [1, 2, 3, 8])
def test_thread_counts_and_read_sizes(check, tmp_path, threads):
    data = text(30000, 2)
    f = tmp_path / "a.gz"
    f.write_bytes(gzip.compress(data, 6))
    for read_size in (None, 1 << 16, 1 << 24):
        rc, line, st = run(check, f, threads, 50000, read_size)
        assert rc == 0, line
        assert st["fell_back"] == "0", line


@pytest.mark.parametrize("what", ["stored", "fixed", "huffman_only", "rle", "sync_flush", "full_flush", "tiny_window"])
def test_block_flavours(check, tmp_path, what):
    data = text(20000, 3)
    kw = {"stored": dict(level=0), "fixed": dict(strategy=zlib.Z_FIXED), "huffman_only": dict(strategy=zlib.Z_HUFFMAN_ONLY),
          "rle": dict(strategy=zlib.Z_RLE), "sync_flush": dict(flush_every=70000), "tiny_window": dict(mem=1),
          "full_flush": dict(flush_every=50000, flush=zlib.Z_FULL_FLUSH)}[what]
    f = tmp_path / "a.gz"
    f.write_bytes(member(data, **kw))
    for chunk in (20000, 300000):
        rc, line, st = run(check, f, 4, chunk)
        assert rc == 0, line
        assert int(st["bytes"]) == len(data)


def test_many_members_and_what_lies_between(check, tmp_path):
    r = random.Random(7)
    data = text(30000, 4)
    parts, o = [], 0
    while o < len(data):
        n = r.choice([0, 1, 50, 5000, 70000, 400000])
        parts.append(data[o:o + n])
        o += n
    raw = b"".join(member(p, level=r.choice([0, 1, 6])) for p in parts)
    f = tmp_path / "m.gz"
    f.write_bytes(raw)
    for chunk in (8192, 60000, 1 << 20):
        rc, line, st = run(check, f, 4, chunk)
        assert rc == 0, line
        assert int(st["bytes"]) == len(data)
        assert int(st["members"]) == len(parts), line
    # members of 1 MiB, as the programs' own gzip output is made (fq_parallel.h: GzipMembers)
    raw = b"".join(member(data[o:o + (1 << 20)], level=4) for o in range(0, len(data), 1 << 20))
    f.write_bytes(raw)
    rc, line, st = run(check, f, 4, 150000)
    assert rc == 0 and st["fell_back"] == "0", line


@pytest.mark.parametrize("tail", [b"\x00", b"\x1f", b"garbage behind the last member\n" * 100, b"\x1f\x8b", b"\x1f\x8b\x08\x00\x00",
                                  b"\x1f\x8b\x07\x00\x00\x00\x00\x00\x00\x03abc", b"\x1f\x8b\x08\xe0\x00\x00\x00\x00\x00\x03abc"])
def test_bytes_behind_the_last_member(check, tmp_path, tail):
    data = text(8000, 5)
    f = tmp_path / "t.gz"
    f.write_bytes(member(data) + tail)
    for chunk in (30000, 1 << 22):
        rc, line, st = run(check, f, 4, chunk)
        assert rc == 0, line


def test_header_fields(check, tmp_path):
    data = text(8000, 6)
    hdr = (b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\x00\x00\x00\x00\x00\x03" + struct.pack("<H", 9) + b"AB\x05\x00hello" +
           b"file name.fastq\x00" + b"a comment\x00")
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    f = tmp_path / "h.gz"
    f.write_bytes(member(data[:100000], header=hdr) + member(data[100000:], header=hdr))
    for chunk in (30000, 1 << 22):
        rc, line, st = run(check, f, 4, chunk)
        assert rc == 0, line
        assert int(st["bytes"]) == len(data)


def test_cut_files_end_where_zlib_ends_them(check, tmp_path):
    # (gzread hands out what it could inflate and then says "end of data": Z_BUF_ERROR is no error to it)
    data = text(12000, 8)
    raw = gzip.compress(data, 6)
    f = tmp_path / "c.gz"
    r = random.Random(3)
    for cut in [len(raw) - 1, len(raw) - 4, len(raw) - 8, len(raw) - 9, len(raw) // 2, 100, 18, 11] + [r.randrange(20, len(raw)) for _ in range(6)]:
        f.write_bytes(raw[:cut])
        for chunk in (40000, 1 << 22):
            rc, line, st = run(check, f, 4, chunk)
            assert rc == 0, (cut, line)
            assert int(st["bytes"]) <= len(data), (cut, line)
            assert st["zlib_error"] == "-" and st["error"] == "-", (cut, line)


def test_damaged_files(check, tmp_path):
    data = text(12000, 9)
    raw = bytearray(gzip.compress(data, 6))
    f = tmp_path / "d.gz"
    r = random.Random(11)
    refused = 0
    for k in range(40):
        b = bytearray(raw)
        at = r.randrange(10, len(b))
        b[at] ^= 1 << r.randrange(8)
        f.write_bytes(bytes(b))
        rc, line, st = run(check, f, 4, r.choice([20000, 90000, 1 << 22]))
        assert rc == 0, (at, line)
        refused += st["zlib_error"] != "-"
    assert refused >= 35  # (a flipped bit in a stored length or in the trailer is always noticed; nearly all others too)


def test_other_kinds_of_content(check, tmp_path):
    r = random.Random(13)
    cases = {
        "empty": b"",
        "one_byte": b"@",
        "noise": bytes(r.getrandbits(8) for _ in range(600000)),
        "zeros": bytes(3000000),
        "repeats": (b"ACGT" * 37 + b"\n") * 40000,
        "mixed": text(4000, 10) + bytes(r.getrandbits(8) for _ in range(200000)) + text(4000, 11),
    }
    f = tmp_path / "o.gz"
    for name, data in cases.items():
        for level in (1, 9):
            f.write_bytes(gzip.compress(data, level))
            for chunk in (4096, 50000):
                rc, line, st = run(check, f, 4, chunk)
                assert rc == 0, (name, line)
                assert int(st["bytes"]) == len(data), (name, line)


def test_a_gzip_file_inside_a_gzip_file(check, tmp_path):
    """the content is itself deflate data: stored in the outer member byte for byte, so the search for block starts finds
    the INNER file's blocks - guesses that decode for a long while and are wrong for the outer stream.  They must cost
    time only."""
    inner = gzip.compress(text(40000, 15), 6)
    f = tmp_path / "g.gz"
    for outer in (member(inner, level=0), member(inner, level=1), member(inner + text(3000, 16) + inner, level=6)):
        f.write_bytes(outer)
        for chunk in (30000, 200000):
            rc, line, st = run(check, f, 4, chunk)
            assert rc == 0, line
            assert st["zlib_error"] == "-" and st["error"] == "-", line


def test_files_that_are_no_gzip_files_come_out_as_they_are(check, tmp_path):
    # (zlib's gzread is transparent for them; the programs never send such a file here - they look at the magic first)
    data = text(3000, 14)
    raw = bytearray(gzip.compress(data, 6))
    raw[1] ^= 0x10
    f = tmp_path / "p"
    for blob in (b"plain text\n" * 100000, b"\x1f", b"", bytes(raw), data):
        f.write_bytes(blob)
        rc, line, st = run(check, f, 4, 30000)
        assert rc == 0 and int(st["bytes"]) == len(blob), line


def test_a_reference_in_front_of_the_members_first_byte(check, tmp_path):
    # a second member whose first block copies from "before the member": zlib says "invalid distance too far back"
    data = text(3000, 12)
    good = member(data)
    # raw deflate of a block that starts with a match: compress with a preset dictionary, drop the dictionary
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, data[-20000:])
    body = c.compress(data[-20000:] + data[:5000]) + c.flush()
    bad = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body + struct.pack("<II", 0, 25000)
    f = tmp_path / "x.gz"
    for raw in (good + bad, bad):
        f.write_bytes(raw)
        for chunk in (10000, 1 << 22):
            rc, line, st = run(check, f, 4, chunk)
            assert rc == 0, line
            assert "too far back" in st.get("zlib_error", "") or "too far back" in line, line


def stored_member(payload):
    """a member of ONE stored block: its compressed length is len(payload) + 10 (header) + 5 (block) + 8 (trailer)"""
    assert len(payload) < 65536
    body = b"\x01" + struct.pack("<HH", len(payload), len(payload) ^ 0xFFFF) + payload
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


@pytest.mark.parametrize("threads", [1, 3, 4])
@pytest.mark.parametrize("lead", [0, 3], ids=["first_round", "later_round"])
def test_a_member_that_ends_where_the_loaded_window_ends(check, tmp_path, threads, lead):
    """The trailer of a member at (or one byte before, or behind) the end of the bytes a round has loaded, with more
    members behind it: 0 or 1 bytes in the middle of a file decide nothing about what follows (round 4 took them for
    bytes behind the last member and ended the file there, silently)."""
    chunk = 4096
    r = random.Random(11 + threads)
    rest = [r.randbytes(r.choice([1, 700, 3000])) for _ in range(3)]
    before = [r.randbytes(chunk * threads + 17) for _ in range(lead)]  # whole windows of members in front of it
    f = tmp_path / "w.gz"
    window = chunk * threads
    for total in range(window - 14, window + 40):  # the compressed length of the member in question
        payload = r.randbytes(total - 23)
        parts = before + [payload] + rest
        f.write_bytes(b"".join(stored_member(p) if len(p) < 65536 else member(p, level=0) for p in parts))
        rc, line, st = run(check, f, threads, chunk)
        assert rc == 0, (total, line)
        assert int(st["bytes"]) == sum(len(p) for p in parts), (total, line)
        assert int(st["members"]) == len(parts) and st["error"] == "-", (total, line)


def test_a_header_cut_by_the_window_end(check, tmp_path):
    """... and a next member whose header (with a name and an extra field) straddles the end of the loaded window"""
    chunk, threads = 4096, 2
    r = random.Random(5)
    header = b"\x1f\x8b\x08\x0c\x00\x00\x00\x00\x00\x03" + struct.pack("<H", 6) + b"abcdef" + b"a name.fastq\x00"
    f = tmp_path / "h.gz"
    for total in range(chunk * threads - 40, chunk * threads + 12):
        a, b = r.randbytes(total - 23), r.randbytes(900)
        f.write_bytes(stored_member(a) + member(b, header=header, level=6) + stored_member(b"tail"))
        rc, line, st = run(check, f, threads, chunk)
        assert rc == 0 and int(st["bytes"]) == len(a) + len(b) + 4 and int(st["members"]) == 3, (total, line)
