"""bam_add_tags on the GPU: the drop-in program bin/bam_add_tags against the golden invocations of the
reference binary (tests/golden/bam_tags.json: exit status, stderr, the INFLATED output BAM), and the bulk call
fqg_bam_add_tags through the C-ABI against the oracle (oracle/bam_tags_oracle.py) on seeded streams: names that leave
get_barcodes at every exit, records from a few bytes to tens of kilobytes (tiles that fit LDS and tiles that do not),
--10x / --tx / --tx_2_gx, device-resident input, and the inputs this build refuses."""
import gzip
import hashlib
import json
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import bam_tags_oracle as bto
from tests import bamgen
from tests.util import GOLD, REPO, SideBySide

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "bin", "bam_add_tags")
GOLDEN = json.load(open(os.path.join(GOLD, "bam_tags.json")))


def golden_run(i):
    case = GOLDEN[i]
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        args = [rel + "/o.bam" if a == "OUT" else a for a in case["args"]]
        p = subprocess.run(["bam_add_tags"] + args, executable=BIN, cwd=GOLD, capture_output=True, timeout=300)
        path = os.path.join(tmp, "o.bam")
        written = open(path, "rb").read() if os.path.exists(path) else None
    return p.returncode, p.stdout, p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), written


# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
GOLDEN_RUNS = SideBySide(golden_run, range(len(GOLDEN)))


@pytest.mark.parametrize("i", range(len(GOLDEN)), ids=[" ".join(c["args"])[-70:] or "no arguments" for c in GOLDEN])
def test_golden_invocations(i):
    case = GOLDEN[i]
    rc, out, err, written = GOLDEN_RUNS.get(i)
    assert rc == case["exit"], err[-500:]
    assert err == case["stderr"]
    if not case["stdout_is_bam"]:
        assert (written is not None) == case["out_created"]
        assert out.decode("latin-1") == case.get("stdout", "")
    if "out_sha256" in case:
        data = gzip.decompress(out if case["stdout_is_bam"] else written)
        assert len(data) == case["out_bytes"]
        assert hashlib.sha256(data).hexdigest() == case["out_sha256"]


@pytest.fixture(scope="module")
def ctx():
    import fastq_utils_amd as fq
    c = fq.Context(0)
    yield c
    c.close()


def make_stream(rng, n, long_every=0, refs=12):
    names = [b"TX%04d.%d" % (i, i % 3) for i in range(refs)]
    recs = []
    for i in range(n):
        c = bamgen.barcode(rng, int(rng.choice([0, 8, 16, 49])))
        u = bamgen.barcode(rng, int(rng.choice([0, 10, 12])))
        s = bamgen.barcode(rng, int(rng.choice([0, 0, 8])))
        tail = b"M%d:%d" % (i, int(rng.integers(0, 10 ** 6)))
        kind = int(rng.integers(0, 10))
        if kind < 6:
            name = b"STAGS_CELL=%s_UMI=%s_SAMPLE=%s_ETAGS_%s" % (c, u, s, tail)
        elif kind == 6:
            name = tail
        elif kind == 7:
            name = b"STAGS_CELL=%s_UMX=%s_SAMPLE=%s_ETAGS_%s" % (c, u, s, tail)
        elif kind == 8:
            name = b"STAGS_CELL=%s_UMI=%s_SAMPLE=%s_" % (c, u, s)
        else:
            name = b"STAGS_" + tail + b"_"
        seq_len = int(rng.integers(0, 90))
        if long_every and i % long_every == long_every - 1:
            seq_len = int(rng.integers(8000, 30000))
        aux = bamgen.aux_z(b"XA", b"x" * int(rng.integers(0, 20))) if rng.random() < 0.5 else b""
        tid = -1 if rng.random() < 0.15 else int(rng.integers(0, refs))
        recs.append(bamgen.record(name[:254], aux, tid=tid, seq_len=seq_len))
    return bamgen.header(tuple((nm, 1000) for nm in names)) + b"".join(recs), names


def oracle_records(stream, **kw):
    out, n = bto.add_tags_stream(stream, **kw)
    _, first = bto.parse_header(stream)
    return out[first:], n


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", ["plain", "10x", "tx", "tx_gx"])
def test_bulk_call_against_the_oracle(ctx, seed, mode):
    rng = np.random.default_rng(100 * seed + len(mode))
    n = int(rng.choice([1, 63, 64, 65, 700, 5000]))
    stream, names = make_stream(rng, n, long_every=[0, 0, 97, 5][seed % 4])
    genes = {nm: b"GENE_%d" % (i // 2) for i, nm in enumerate(names) if i % 4 != 3}
    kw = {"plain": {}, "10x": {"tenx": True}, "tx": {"tx_tag": True}, "tx_gx": {"tx_tag": True, "tmap": genes}}[mode]
    want, n_aln = oracle_records(stream, **kw)
    got = ctx.bam_add_tags(stream, tenx=kw.get("tenx", False), tx_tag=kw.get("tx_tag", False), targets=names,
                           genes=kw.get("tmap"))
    assert got["code"] == 0 and got["n_alignments"] == n_aln == n
    assert got["out_bytes"] == len(want)
    assert got["records"] == want


def test_device_resident_stream(ctx):
    import torch
    rng = np.random.default_rng(9)
    stream, names = make_stream(rng, 3000, long_every=211)
    want, n = oracle_records(stream, tx_tag=True)
    pad = 5  # a stream that does not start at a 16-byte boundary is copied first
    for shift in (0, pad):
        t = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda:0")
        t[shift:shift + len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).to("cuda:0")
        offs = []
        p = bto.parse_header(stream)[1]
        while p + 4 <= len(stream):
            offs.append(p)
            p += 4 + struct.unpack_from("<i", stream, p)[0]
        got = ctx.bam_add_tags(t.data_ptr() + shift, tx_tag=True, targets=names, offsets=offs, nbytes=len(stream))
        assert got["code"] == 0 and got["records"] == want


def test_names_the_reference_has_no_defined_output_for(ctx):
    """A value that runs to the end of the record (no '_' behind it), or is 50 characters or longer: the reference
    reads behind the record / writes behind its arrays; the bulk call reports the first such alignment instead."""
    ok = bamgen.record(b"STAGS_CELL=AC_UMI=GT_SAMPLE=_ETAGS_r", b"")
    hdr = bamgen.header(((b"chr1", 10),))
    run_on = bamgen.record(b"STAGS_CELL=ACGT", b"", seq_len=0)   # no '_' in what follows (cigar and the rest are empty or not '_')
    got = ctx.bam_add_tags(hdr + ok + ok + run_on + ok, targets=[b"chr1"])
    with pytest.raises(bto.Undefined):
        bto.add_tags_stream(hdr + run_on)
    assert got["code"] == 22  # FQG_E_TAGS_NAME
    assert got["record"] == 2
    long_cell = bamgen.record(b"STAGS_CELL=" + b"A" * 50 + b"_UMI=_SAMPLE=_", b"")
    got = ctx.bam_add_tags(hdr + long_cell, targets=[b"chr1"])
    assert got["code"] == 22 and got["record"] == 0
    fine = bamgen.record(b"STAGS_CELL=" + b"A" * 49 + b"_UMI=_SAMPLE=_", b"")
    got = ctx.bam_add_tags(hdr + fine, targets=[b"chr1"])
    assert got["code"] == 0 and got["records"] == oracle_records(hdr + fine)[0]
    beyond = bamgen.record(b"STAGS_CELL=A_UMI=C_SAMPLE=_x", b"", tid=3)
    got = ctx.bam_add_tags(hdr + beyond, tx_tag=True, targets=[b"chr1"])
    assert got["code"] == 23 and got["record"] == 0
    got = ctx.bam_add_tags(hdr + beyond, tx_tag=False, targets=[b"chr1"])   # without --tx the id is never looked at
    assert got["code"] == 0


def test_empty_input(ctx):
    hdr = bamgen.header(((b"chr1", 10),))
    got = ctx.bam_add_tags(hdr, targets=[b"chr1"])
    assert got["code"] == 0 and got["n_alignments"] == 0 and got["out_bytes"] == 0
