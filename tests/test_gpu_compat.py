"""The per-record C API of the reference (fastq.h / hash.h) on the GPU library: the reference's OWN
fastq_info.c - compiled unmodified against the reference's own headers in the build container
(oracle/Makefile) and linked with libfastq_gpu.so instead of fastq.o + hash.o - must behave like the
reference binary on every golden invocation (tests/golden/fastq_info.json: exit status, stdout,
stderr).  The binary travels to the GPU box prebuilt; nothing of the reference is read here."""
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
