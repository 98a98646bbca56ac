"""Host-side (de)compression on all cores (fastq_utils_amd/host/fq_parallel.h): what a
consumer inflates must be exactly the bytes that went in.  CPU only; compiles a small driver."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from tests import bamgen
from tests.util import REPO

SRC = os.path.join(REPO, "tests", "cxx", "host_parallel_check.cpp")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hp") / "host_parallel_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, SRC, "-lz"], check=True)
    return exe


@pytest.mark.parametrize("threads", ["1", "3", "8"])
@pytest.mark.parametrize("size", [0, 1, 70_000, 9_500_000])
def test_gzip_members_inflate_to_the_input(driver, tmp_path, threads, size):
    rng = np.random.default_rng(size + 1)
    data = bytes(rng.choice(np.frombuffer(b"ACGTN\n@+IIFF#", dtype=np.uint8), size).astype(np.uint8))
    src, dst = tmp_path / "in.txt", tmp_path / "out.gz"
    src.write_bytes(data)
    env = dict(os.environ, FQGPU_HOST_THREADS=threads)
    subprocess.run([driver, "gz", str(src), str(dst), "4"], check=True, env=env)
    raw = dst.read_bytes()
    assert gzip.decompress(raw) == data  # python reads through members
    # zlib's gzread does too (what the reference's own readers use): inflate member after member
    out, rest = b"", raw
    while rest:
        d = zlib.decompressobj(31)
        out += d.decompress(rest)
        rest = d.unused_data
    assert out == data
    if size > 8_388_608:
        assert raw.count(b"\x1f\x8b\x08\x00") >= 3  # more than one member


@pytest.mark.parametrize("threads", ["1", "5"])
def test_bgzf_blocks_inflate_in_parallel(driver, tmp_path, threads):
    rng = np.random.default_rng(5)
    bam, _ = bamgen.tagged_bam(rng, n_cells=40, genes=80, fresh_umis=True)
    src, dst = tmp_path / "in.bam", tmp_path / "out.bin"
    src.write_bytes(bam)
    env = dict(os.environ, FQGPU_HOST_THREADS=threads)
    subprocess.run([driver, "bgzf", str(src), str(dst)], check=True, env=env)
    assert dst.read_bytes() == gzip.decompress(bam)


def test_bgzf_rejects_a_plain_gzip_file(driver, tmp_path):
    src = tmp_path / "x.gz"
    src.write_bytes(gzip.compress(b"not bgzf"))
    assert subprocess.run([driver, "bgzf", str(src), str(tmp_path / "o")]).returncode == 7


@pytest.mark.parametrize("threads", ["1", "6"])
@pytest.mark.parametrize("size,split", [(0, 0), (1, 0), (65280, 65280), (65281, 10), (3_000_000, 1_234_567)])
def test_bgzf_writer_makes_a_file_every_bgzf_reader_takes(driver, tmp_path, threads, size, split):
    """bgzf_deflate_parallel (the output side of bin/bam_add_tags): blocks of at most 64 KiB with the BC field, the
    empty end-of-file block last; inflating gives the two pieces back to back - with python's gzip, with the oracle's
    block walker, and with this repo's own parallel reader."""
    from oracle.umi_oracle import bgzf_inflate
    rng = np.random.default_rng(size + 3)
    data = bytes(rng.integers(0, 256, size, dtype=np.uint8)) if size < 100000 else \
        bytes(rng.choice(np.frombuffer(b"ACGT\0\x01\x02STAGS_", dtype=np.uint8), size).astype(np.uint8))
    src, dst, back = tmp_path / "in.bin", tmp_path / "out.bgzf", tmp_path / "back.bin"
    src.write_bytes(data)
    env = dict(os.environ, FQGPU_HOST_THREADS=threads)
    subprocess.run([driver, "tobgzf", str(src), str(dst), str(split)], check=True, env=env)
    raw = dst.read_bytes()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    assert gzip.decompress(raw) == data
    assert bgzf_inflate(raw) == data
    subprocess.run([driver, "bgzf", str(dst), str(back)], check=True, env=env)
    assert back.read_bytes() == data
