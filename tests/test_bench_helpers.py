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


def _old_line():
    """a complete bench document of an earlier round (22 KB as ONE line: the driver could not read it)"""
    import json

    with open(os.path.join(REPO, "profiles", "r04c_bench_100M.json")) as f:
        return json.load(f)


def test_final_line_is_small_and_parses():
    import json

    b = load_bench()
    doc = _old_line()
    head = {k: doc[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "host_fed")}
    extras = {k: v for k, v in doc.items() if k.endswith("_extra") or k == "e2e"}
    # what sank round 4: a captured stderr full of backspaces, and anything else a program may say
    extras["filterpair_extra"]["program"]["says"] = "\b" * 150 + "caf\xe9 \x00\x1b[0m\n\t "
    head["cpu_baseline"]["sample"] += "\n\b\x7f"
    for final, ex in ((False, None), (True, extras)):
        line = b.headline_line(head, ex, final=final)
        assert len(line) <= 4096 and len(line.encode()) == len(line)
        assert all(32 <= ord(c) < 127 for c in line), [c for c in line if not 32 <= ord(c) < 127]
        assert "\\b" not in line and "\\u" not in line and "\\n" not in line
        got = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "final"):
            assert k in got, k
        assert got["final"] is final and got["config"]["workload"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                  "algorithmic_bytes_per_launch", "avg_launch_ms", "all_kernels_ms_per_step"):
            assert k in got["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample", "seconds", "ok"):
            assert k in got["cpu_baseline"], k
        assert abs(got["value"] - doc["value"]) <= 1e-5 * doc["value"]
        assert abs(got["roofline"]["frac"] - doc["roofline"]["frac"]) <= 1e-5
        if final:
            assert got["extras_ok"] is True and got["extras_failed"] == []
            assert got["host_fed"]["abi_Mreads_per_s"] and got["extras"]["umi_count_kernels_ms"]
    # every extra is a line of its own, printable ASCII as well
    for name, block in extras.items():
        text = b.extra_line(name, block)
        assert all(32 <= ord(c) < 127 for c in text) and "\\b" not in text
        assert json.loads(text)["extra"] == name


def test_a_failed_comparison_in_an_extra_shows_in_the_line():
    import json

    b = load_bench()
    doc = _old_line()
    extras = {k: v for k, v in doc.items() if k.endswith("_extra")}
    extras["umi_count_extra"]["matrix_identical_to_reference_program"] = False
    extras["filters_extra"] = {"error": "RuntimeError('x')"}
    got = json.loads(b.headline_line(doc, extras))
    assert got["extras_ok"] is False and sorted(got["extras_failed"]) == ["filters_extra", "umi_count_extra"]


def test_the_line_sheds_optional_blocks_before_it_grows_past_the_limit():
    import json

    b = load_bench()
    doc = _old_line()
    doc["host_fed"]["what"] = "x" * 5000
    got = json.loads(b.headline_line(doc, {}))
    assert "host_fed" not in got and "roofline" in got and "cpu_baseline" in got
