"""The fast deflate compressor behind FQGPU_GZIP_FAST (fastq_utils_amd/host/fq_fastdeflate.h): whatever it is given, zlib
must inflate its output back to the input.  CPU only; tests/cxx/fastdeflate_check.cpp does the round trip."""
import os
import random
import re
import subprocess

import pytest

from tests.test_pgzip import fastq_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "tests", "cxx")


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    d = tmp_path_factory.mktemp("fdef")
    path = str(d / "fastdeflate_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", path,
                    os.path.join(CXX, "fastdeflate_check.cpp"), "-lz"], check=True)
    return path


def run(check, path, member):
    p = subprocess.run([check, str(path), str(member)], capture_output=True, text=True, timeout=600)
    st = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", p.stdout)}
    return p.returncode, p.stdout.strip(), st


def contents():
    r = random.Random(5)
    yield "empty", b""
    yield "one_byte", b"@"
    yield "seven_bytes", b"ACGTACG"
    yield "fastq", fastq_text(12000, 1)
    yield "fastq_long_reads", fastq_text(300, 2, read_len=(3000, 9000))
    yield "noise", r.randbytes(700000)
    yield "zeros", bytes(2000000)
    yield "runs", b"".join(bytes([r.randrange(256)]) * r.randrange(1, 2000) for _ in range(3000))
    yield "period_32768", (r.randbytes(32768) * 6)[:190000]   # matches at the window's full distance
    yield "period_32769", (r.randbytes(32769) * 6)[:190000]   # ... and just beyond it
    yield "text", b"".join(r.choice([b"the ", b"quick ", b"brown ", b"fox\n", b"jumps ", b"over "]) for _ in range(200000))
    yield "two_symbols", bytes(r.choice(b"AB") for _ in range(300000))
    yield "skewed", bytes(min(255, int(r.expovariate(0.02))) for _ in range(400000))   # deep Huffman trees: the length limit
    yield "mixed", fastq_text(3000, 3) + r.randbytes(100000) + bytes(50000) + fastq_text(3000, 4)


@pytest.mark.parametrize("name,data", list(contents()), ids=[n for n, _ in contents()])
def test_round_trip_through_zlib(check, tmp_path, name, data):
    f = tmp_path / "in"
    f.write_bytes(data)
    for member in (1 << 20, 70001, 4096):
        if member == 4096 and len(data) > 1000000:
            continue
        rc, line, st = run(check, f, member)
        assert rc == 0 and st["in"] == len(data), (name, member, line)
        if name in ("fastq", "text", "zeros", "runs", "period_32768") and member == 1 << 20:
            assert st["out"] < 0.6 * len(data), (name, line)  # it does compress
        if name == "noise":
            assert st["out"] < 1.01 * len(data) + 200 * st["members"], line  # and does not blow noise up


def test_random_structures(check, tmp_path):
    f = tmp_path / "in"
    for seed in range(40):
        r = random.Random(seed)
        parts = []
        for _ in range(r.randrange(1, 12)):
            kind = r.randrange(5)
            n = r.choice([1, 7, 100, 5000, 90000])
            if kind == 0:
                parts.append(r.randbytes(n))
            elif kind == 1:
                parts.append(bytes([r.randrange(256)]) * n)
            elif kind == 2:
                parts.append((r.randbytes(r.randrange(1, 40)) * (n // 3 + 1))[:n])
            elif kind == 3:
                parts.append(fastq_text(n // 300 + 1, seed)[:n])
            else:
                parts.append(b"".join(parts)[-n:])  # a copy of what came before
        data = b"".join(parts)
        f.write_bytes(data)
        rc, line, st = run(check, f, r.choice([1 << 20, 33333, 1000]))
        assert rc == 0 and st["in"] == len(data), (seed, line)
