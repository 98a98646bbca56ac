"""The per-record C API of the reference (fastq.h / hash.h) on the GPU library: the reference's OWN
fastq_info.c and fastq_filterpair.c - compiled unmodified against the reference's own headers in the build
container (oracle/Makefile) and linked with libfastq_gpu.so instead of fastq.o + hash.o - must behave like the
reference binaries on every golden invocation (tests/golden/fastq_info.json, filterpair.json: exit status,
stdout, stderr, decompressed outputs).  The binaries travel to the GPU box prebuilt; nothing of the reference is
read here."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from tests.util import GOLD, REPO, load_fastq_info_golden, strip_progress

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu")
LIB = os.path.join(REPO, "fastq_utils_amd", "libfastq_gpu.so")
GOLDEN = load_fastq_info_golden()


def run(args):
    p = subprocess.run([BIN] + args, cwd=GOLD, capture_output=True, timeout=300)
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


def test_reference_main_program_on_the_gpu_library():
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/fastq_info_on_libfastq_gpu was not built (needs the reference checkout at build time)")
    assert os.path.exists(LIB)

    def one(case):
        rc, out, err = run(case["args"])
        ok = (rc == case["exit"] and out == case["stdout"]
              and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], rc, case["exit"], out[-200:], case["stdout"][-200:], err[-500:],
                                case["stderr"][-500:])

    with ThreadPoolExecutor(8) as ex:
        bad = [b for b in ex.map(one, GOLDEN) if b]
    assert not bad, f"{len(bad)} of {len(GOLDEN)} differ; first: {bad[:4]}"


def test_reference_filterpair_program_on_the_gpu_library():
    """src/fastq_filterpair.c (run_tests.sh:361-370) uses what fastq_info does not: fastq_rewind,
    fastq_quick_copy_entry, lookups of two files in each other's index, a file's lookups in its own index."""
    import gzip
    import hashlib
    import json
    import tempfile

    exe = os.path.join(REPO, "oracle", "_ref", "fastq_filterpair_on_libfastq_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/fastq_filterpair_on_libfastq_gpu was not built")
    golden = json.load(open(os.path.join(GOLD, "filterpair.json")))

    def one(case):
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = list(case["args"])
            if len(real) in (2, 3):
                real = real[:2] + [rel + "/p1.fastq.gz", rel + "/p2.fastq.gz", rel + "/up.fastq.gz"] + real[2:]
            real = [rel + "/" + a if a in ("O1", "O2") else a for a in real]
            p = subprocess.run(["fastq_filterpair"] + real, executable=exe, cwd=GOLD, capture_output=True, timeout=300)
            err = p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/")
            if (p.returncode, p.stdout.decode("latin-1"), strip_progress(err)) != (
                    case["exit"], case["stdout"], strip_progress(case["stderr"])):
                return case["args"], p.returncode, case["exit"], err[-400:], case["stderr"][-400:]
            for k, want in case["files"].items():
                got = gzip.decompress(open(os.path.join(tmp, k + ".fastq.gz"), "rb").read())
                if hashlib.sha256(got).hexdigest() != want["sha256"]:
                    return case["args"], "file", k
        return None

    with ThreadPoolExecutor(4) as ex:
        bad = [b for b in ex.map(one, golden) if b]
    assert not bad, f"{len(bad)} of {len(golden)} differ; first: {bad[:3]}"
