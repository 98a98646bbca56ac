"""The reference's line reader as a cut list (fastq_utils_amd/host/fq_reframe.h, the re-framing mode of
host/fq_input.h): what the consumer of a re-framed input sees must be the pieces the reference's gzgets calls return
(src/fastq.c:249-253: limits of 1000 bytes for the header lines, 2 500 000 for sequence and quality, one call per field
of a record), each piece that comes back without its newline followed by "\\0\\n".  CPU only."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests.util import REPO

CXX = os.path.join(REPO, "tests", "cxx")
LIMITS = (999, 2_499_999)


def gzgets_image(data):
    """the re-framed image by the definition: walk the bytes with the reference's four calls per record"""
    out, pos, ph = [], 0, 0
    while pos < len(data):
        lim = LIMITS[ph & 1]
        k = data.find(b"\n", pos, pos + lim)  # (no copy of the 2.5 MB a sequence call may look at)
        piece = data[pos:pos + lim] if k < 0 else data[pos:k + 1]
        pos += len(piece)
        ph += 1
        if piece.endswith(b"\n") or (len(piece) < lim and pos == len(data)):
            out.append(piece)  # a whole line, or the unterminated rest of the file
        else:
            out.append(piece + b"\0\n")
    return b"".join(out)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("reframe")
    path = str(d / "reframe_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-o", path, os.path.join(CXX, "reframe_check.cpp"), "-lz"], check=True)
    return path


def images():
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rec(i, hdr_extra=0, read_len=50, hdr2_extra=0):
        s = acgt[rng.integers(0, 4, read_len)].tobytes()
        return (b"@r%d" % i + b"h" * hdr_extra + b"\n" + s + b"\n+" + b"x" * hdr2_extra + b"\n" + b"I" * read_len + b"\n")

    ok = b"".join(rec(i) for i in range(40))
    out = {
        "clean": ok,
        "header_998": rec(0, 995) + ok,                # a header line of exactly limit bytes with its newline: no cut
        "header_999": rec(0, 996) + ok,                # one byte more: the newline alone is the next piece
        "header_1500": ok + rec(1, 1500) + ok,
        "header_5000": rec(2, 5000) + ok,              # several cuts in one line, alternating limits
        "hdr2_1200": ok + rec(3, 0, 50, 1200) + ok,
        "read_2.6M": ok + rec(4, 0, 2_600_000) + ok,
        "read_5.1M": rec(5, 0, 5_100_000) + ok,
        "no_final_newline": ok + rec(6, 1500)[:-1],
        "one_long_line_no_newline": b"@" + b"q" * 7000,
        "only_newlines": b"\n" * 5000,
        "ends_at_a_cut": b"@" + b"z" * 998,            # the file ends exactly where a cut would fall
        "nul_bytes": ok + b"@a\0b" + b"h" * 1200 + b"\nAC\0GT\n+\nII\0II\n" + ok,
    }
    many = []
    for i in range(300):
        many.append(rec(i, int(rng.integers(0, 1300)) if i % 7 == 0 else 0, int(rng.integers(1, 300)), 1100 if i % 31 == 5 else 0))
    out["many_cuts"] = b"".join(many)
    # Megabytes of ordinary records in front of the long lines: pieces of them take the short cut of the stager (a search
    # for the longest line on many threads, host/fq_input.h: short_lines_only), which must leave the line reader's state
    # as the walk line by line would - the stray line puts every later line one call out of step
    plain = b"".join(rec(i) for i in range(60000))
    out["long_lines_behind_7MB"] = plain + b"stray line\n" + rec(7, 1500) + ok + rec(8, 0, 2_600_000) + ok
    out["7MB_without_a_long_line"] = plain + b"stray line\n" + ok
    return out


IMAGES = images()


@pytest.mark.parametrize("mode", ["pieces_64k", "pieces_5M", "whole", "gz_pieces"])
@pytest.mark.parametrize("name", sorted(IMAGES))
def test_consumer_sees_the_gzgets_pieces(exe, tmp_path, name, mode):
    data = IMAGES[name]
    gz = mode == "gz_pieces"
    src = tmp_path / ("in.fastq.gz" if gz else "in.fastq")
    src.write_bytes(gzip.compress(data, 1) if gz else data)
    piece = {"pieces_64k": 65536, "pieces_5M": 5 << 20, "whole": 1 << 20, "gz_pieces": 300_000}[mode]
    env = dict(os.environ)
    if gz:
        env.pop("FQGPU_REFRAME", None)  # inflated input is always cut as it is read
    else:
        env["FQGPU_REFRAME"] = "1"
    p = subprocess.run([exe, str(src), str(piece), "2" if mode == "whole" else "0"], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    want = gzgets_image(data)
    assert len(p.stdout) == len(want) and p.stdout == want
    if name == "clean":
        assert want == data
