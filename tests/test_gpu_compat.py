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

from tests.util import GOLD, REPO, load_fastq_info_golden, strip_progress, thinned

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu")
LIB = os.path.join(REPO, "fastq_utils_amd", "libfastq_gpu.so")
GOLDEN = load_fastq_info_golden()


def run(args, env=None):
    p = subprocess.run([BIN] + args, cwd=GOLD, capture_output=True, timeout=300, env=dict(os.environ, **(env or {})))
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


POOL = 12  # programs started side by side (see tests/test_gpu_cli.py)


def test_reference_main_program_on_the_gpu_library():
    """Every golden invocation, once.  In every second case (a checksum of the arguments decides, the OTHER half than in
    tests/test_gpu_cli.py) the library reads the .fastq.gz files of the goldens through the many-core gzip reader,
    host/fq_pgzip.h, in chunks of 4 KiB - files this small are one zlib thread's otherwise."""
    import zlib

    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/fastq_info_on_libfastq_gpu was not built (needs the reference checkout at build time)")
    assert os.path.exists(LIB)
    by_chunks = {"FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "4096", "FQGPU_HOST_THREADS": "3"}

    def one(case):
        chunked = not (zlib.crc32(" ".join(case["args"]).encode()) & 1)
        rc, out, err = run(case["args"], by_chunks if chunked else {})
        ok = (rc == case["exit"] and out == case["stdout"]
              and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], chunked, rc, case["exit"], out[-200:], case["stdout"][-200:], err[-500:],
                                case["stderr"][-500:])

    with ThreadPoolExecutor(POOL) as ex:
        cases = thinned(GOLDEN)  # (every invocation on a box that starts programs at the usual rate: tests/util.py)
        bad = [b for b in ex.map(one, cases) if b]
    assert not bad, f"{len(bad)} of {len(cases)} differ; first: {bad[:4]}"


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

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, golden) if b]
    assert not bad, f"{len(bad)} of {len(golden)} differ; first: {bad[:3]}"


REF_INFO = os.path.join(REPO, "oracle", "_ref", "fastq_info")
REF_FP = os.path.join(REPO, "oracle", "_ref", "fastq_filterpair")


@pytest.mark.skipif(not (os.path.exists(BIN) and os.path.exists(REF_INFO)), reason="oracle/_ref not built")
def test_lines_beyond_the_gzgets_buffers_through_the_per_record_api():
    """The reference's own fastq_info.c on libfastq_gpu.so, on files with lines beyond its gzgets buffers
    (src/fastq.c:249-253): the library cuts the file's bytes the way those calls do (host/fq_reframe.h) before the GPU
    frames them, hands the caller's FASTQ_ENTRY the pieces, and speaks of positions in the FILE.  Exit status, stdout and
    stderr of the reference's own objects - and, for fastq_filterpair (offsets, seeks, copies), the same output files."""
    import gzip
    import tempfile

    from tests.test_oracle_vs_ref_fuzz import overlong_images

    imgs = overlong_images()
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for which, img in sorted(imgs.items()):
            os.mkdir(os.path.join(tmp, which))
            with open(os.path.join(tmp, which, "f.fastq"), "wb") as f:
                f.write(img)
            jobs += [(which, args) for args in (["-r", "f.fastq"], ["f.fastq"], ["f.fastq", "pe"], ["f.fastq", "f.fastq"])]

        def one(job):
            which, args = job
            d = os.path.join(tmp, which)
            want = subprocess.run([REF_INFO] + args, cwd=d, capture_output=True, timeout=300)
            got = subprocess.run([BIN] + args, cwd=d, capture_output=True, timeout=300)
            assert (got.returncode, got.stdout, strip_progress(got.stderr.decode("latin-1"))) == (
                want.returncode, want.stdout, strip_progress(want.stderr.decode("latin-1"))), (which, args, got.stderr[-400:])

        with ThreadPoolExecutor(POOL) as ex:
            list(ex.map(one, jobs))
        exe = os.path.join(REPO, "oracle", "_ref", "fastq_filterpair_on_libfastq_gpu")
        if os.path.exists(exe) and os.path.exists(REF_FP):
            # the accepted image (its pieces are valid records) paired with itself: every record is copied through the offsets
            with open(os.path.join(tmp, "a.fastq"), "wb") as f:
                f.write(imgs["pieces_at_the_read_limit"])
            outs = {}
            for tag, binary in (("ref", REF_FP), ("lib", exe)):
                d = os.path.join(tmp, tag)
                os.mkdir(d)
                p = subprocess.run(["fastq_filterpair", "../a.fastq", "../a.fastq", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"],
                                   executable=binary, cwd=d, capture_output=True, timeout=600)
                files = {}
                for k in ("p1", "p2", "up"):
                    path = os.path.join(d, k + ".fastq.gz")
                    files[k] = gzip.decompress(open(path, "rb").read()) if os.path.exists(path) and os.path.getsize(path) else b""
                outs[tag] = (p.returncode, p.stdout, strip_progress(p.stderr.decode("latin-1")), files)
            assert outs["lib"] == outs["ref"], (outs["lib"][:3], outs["ref"][:3])
