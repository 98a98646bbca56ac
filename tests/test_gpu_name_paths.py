"""The name kernels have two ways to read a header (fqg_index_kernels.hip: 128 bytes into registers,
or byte by byte for long lines / the end of the image / Casava headers without a blank).  Both must
give the SAME canonical name and the SAME hash, or a name stored through one path would not be found
through the other.  Inputs built so that the two copies of a name take different paths, all four modes
of fastq_info against the oracle."""
import os
import tempfile

import numpy as np
import pytest

from tests.test_gpu_cli import compare_with_oracle

pytestmark = pytest.mark.gpu


def rec(name, comment, rng, ln=50):
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    s = bases[rng.integers(0, 4, ln)].tobytes()
    q = (rng.integers(5, 40, ln) + 33).astype(np.uint8).tobytes()
    return b"@" + name + comment + b"\n" + s + b"\n+\n" + q + b"\n"


def casava(i, mate, extra=b""):
    return b"SRX:7:FC9:%d:%d:%d:%d" % (i % 8 + 1, 1000 + i % 97, 2000 + i % 89, i), b" %d:N:0:ACGTAC" % mate + extra


def write(tmp, files):
    for k, v in files.items():
        with open(os.path.join(tmp, k), "wb") as f:
            f.write(v)


@pytest.mark.parametrize("pad", [0, 90, 200])
def test_pairing_when_the_mates_headers_take_different_paths(pad):
    """file 2's comments are `pad` bytes longer: above 127 bytes the lookup runs byte-wise while the
    insert of file 1 ran on registers (and vice versa for the names near the end of each image)"""
    rng = np.random.default_rng(pad + 1)
    n = 3000
    f1 = b"".join(rec(*casava(i, 1), rng) for i in range(n))
    f2 = b"".join(rec(*casava(i, 2, b"X" * pad), rng) for i in range(n))
    with tempfile.TemporaryDirectory() as tmp:
        files = {"a_1.fastq": f1, "a_2.fastq": f2}
        write(tmp, files)
        compare_with_oracle(tmp, ["a_1.fastq", "a_2.fastq"], files, {"FQGPU_CHUNK_MB": "1"})
        compare_with_oracle(tmp, ["-s", "a_1.fastq", "a_2.fastq"], files, {"FQGPU_CHUNK_MB": "1"})


@pytest.mark.parametrize("where", ["last_record", "long_line", "no_blank"])
def test_duplicate_whose_copies_take_different_paths(where):
    rng = np.random.default_rng(len(where))
    n = 2500
    recs = [rec(*casava(i, 1), rng) for i in range(n)]
    name, comment = casava(777, 1)
    if where == "last_record":      # the last 128 bytes of the image are read byte-wise
        recs.append(rec(name, comment, rng, ln=20))
    elif where == "long_line":      # same name, a 300-byte comment
        recs.insert(1500, rec(name, comment + b"Y" * 300, rng))
    else:                           # a Casava file whose repeated header has lost its blank and comment
        recs.insert(1500, rec(name, b"", rng))
    img = b"".join(recs)
    with tempfile.TemporaryDirectory() as tmp:
        files = {"d.fastq": img}
        write(tmp, files)
        compare_with_oracle(tmp, ["d.fastq"], files, {"FQGPU_CHUNK_MB": "1"})


def test_interleaved_mates_with_very_different_header_lengths():
    rng = np.random.default_rng(9)
    img = b"".join(rec(*casava(i, 1), rng) + rec(*casava(i, 2, b"Z" * (140 if i % 3 == 0 else 0)), rng) for i in range(2000))
    with tempfile.TemporaryDirectory() as tmp:
        files = {"i.fastq": img}
        write(tmp, files)
        compare_with_oracle(tmp, ["i.fastq", "pe"], files, {"FQGPU_CHUNK_MB": "1"})
