"""ASan / UBSan (and TSan for the reader thread) builds of the CPU-side C / C++ of this repository (SURVEY section 5,
sanitizer row): the oracle restatements (oracle/fq_oracle.c, rl_oracle.c), the container code of libfastq_gpu.so that
needs no GPU (compat/range_list_compat.cpp), the host-side (de)compression (host/fq_parallel.h) and the host stager
(host/fq_input.h).  Each is compiled with a small driver under tests/cxx/ and must run clean."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLD, REPO, read_image

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
CXX = os.path.join(REPO, "tests", "cxx")


def build(tmp, name, cc, srcs, extra=()):
    exe = str(tmp / name)
    subprocess.run([cc] + SAN + list(extra) + ["-o", exe] + srcs, check=True)
    return exe


@pytest.fixture(scope="module")
def tmpdir(tmp_path_factory):
    return tmp_path_factory.mktemp("san")


def test_oracle_c_restatements_run_clean(tmpdir):
    exe = build(tmpdir, "oracle_san", "gcc", ["-std=gnu11", os.path.join(CXX, "oracle_sanitize.c"),
                                               os.path.join(REPO, "oracle", "fq_oracle.c"), os.path.join(REPO, "oracle", "rl_oracle.c")])
    names = ["test_21_1.fastq.gz", "test_21_2.fastq.gz", "test_e9.fastq.gz", "test_e3.fastq.gz", "test_e5.fastq.gz",
             "casava.1.8i.fastq.gz", "test_solid_1.fastq.gz", "nanopore_rna2.fastq.gz", "test_empty.fastq.gz",
             "c18_10000_1.fastq.gz", "c18_10000_2.fastq.gz", "syn_fp_1.fastq", "syn_fp_2.fastq", "syn_fp_2_noat.fastq"]
    plain = {}
    for n in names:
        p = tmpdir / (n + ".txt")
        p.write_bytes(read_image(os.path.join(GOLD, "data", n)))
        plain[n] = str(p)
    for n in names:
        p = subprocess.run([exe, "info", plain[n]], env=ENV, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
    for a, b in (("test_21_1.fastq.gz", "test_21_2.fastq.gz"), ("c18_10000_1.fastq.gz", "c18_10000_2.fastq.gz")):
        p = subprocess.run([exe, "info", plain[a], plain[b]], env=ENV, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
    for a, b in (("syn_fp_1.fastq", "syn_fp_2.fastq"), ("syn_fp_2.fastq", "syn_fp_1.fastq"), ("syn_fp_1.fastq", "syn_fp_2_noat.fastq"),
                 ("c18_10000_1.fastq.gz", "c18_10000_2.fastq.gz"), ("test_e9.fastq.gz", "test_21_1.fastq.gz")):
        p = subprocess.run([exe, "pair", plain[a], plain[b]], env=ENV, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
    p = subprocess.run([exe, "rl"], env=ENV, capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith(b"rl "), p.stderr.decode()[-2000:]


def test_compat_range_list_runs_clean(tmpdir):
    exe = build(tmpdir, "compat_san", "g++", ["-std=c++17", os.path.join(CXX, "compat_sanitize.cpp"),
                                               os.path.join(REPO, "fastq_utils_amd", "compat", "range_list_compat.cpp")])
    p = subprocess.run([exe], env=ENV, capture_output=True, timeout=300)
    assert p.returncode == 0 and b"members seen" in p.stdout, p.stderr.decode()[-2000:]


def test_host_parallel_compression_runs_clean(tmpdir):
    exe = build(tmpdir, "hp_san", "g++", ["-std=c++17", "-pthread", os.path.join(CXX, "host_parallel_check.cpp")], extra=["-lz"])
    rng = np.random.default_rng(3)
    src = tmpdir / "in.txt"
    src.write_bytes(bytes(rng.choice(np.frombuffer(b"ACGTN\n@+IIFF#", dtype=np.uint8), 3_000_000).astype(np.uint8)))
    dst = tmpdir / "out.gz"
    p = subprocess.run([exe, "gz", str(src), str(dst), "4"], env=dict(ENV, FQGPU_HOST_THREADS="4"), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert gzip.decompress(dst.read_bytes()) == src.read_bytes()


def test_fast_deflate_runs_clean(tmpdir):
    """fq_fastdeflate.h (FQGPU_GZIP_FAST) under ASan + UBSan: the round trip through zlib on text, noise, runs and an empty input"""
    exe = str(tmpdir / "fdef_san")
    subprocess.run(["g++", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-o", exe,
                    os.path.join(CXX, "fastdeflate_check.cpp"), "-lz"], check=True)
    rng = np.random.default_rng(8)
    lines = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(1, 400))).astype(np.uint8)) for _ in range(8000)]
    cases = {"text": b"@r\n".join(lines), "noise": bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), "zeros": bytes(500000), "empty": b"",
             "skewed": bytes(np.minimum(255, rng.exponential(40, 300000)).astype(np.uint8))}
    for name, data in cases.items():
        f = tmpdir / name
        f.write_bytes(data)
        for member in ("1048576", "5000"):
            p = subprocess.run([exe, str(f), member], env=ENV, capture_output=True, timeout=300)
            assert p.returncode == 0 and p.stdout.startswith(b"ok"), (name, member, p.stdout, p.stderr.decode()[-1500:])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_input_stager_runs_clean(tmpdir, san):
    flags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g", "-O1"]
    exe = str(tmpdir / ("input_" + san.split(",")[0]))
    subprocess.run(["g++", "-std=c++17", "-pthread"] + flags + ["-o", exe, os.path.join(CXX, "input_sanitize.cpp"), "-lz"], check=True)
    rng = np.random.default_rng(11)
    lines = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(1, 400))).astype(np.uint8)) for _ in range(20000)]
    data = b"\n".join(lines) + b"\n"
    plain = tmpdir / "x.txt"
    plain.write_bytes(data)
    gz = tmpdir / "x.txt.gz"
    gz.write_bytes(gzip.compress(data, 1))
    empty = tmpdir / "empty.txt"
    empty.write_bytes(b"")

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    want = ("%d %d" % (len(data), fnv(data))).encode()
    # bgzip'd input (BGZF: small gzip members that carry their size) is inflated on many threads (fq_input.h, read_bgzf);
    # FQGPU_NO_PARALLEL_INFLATE sends the same file through zlib's gzread instead - the same bytes either way
    from tests import bamgen
    bgz = tmpdir / "x.txt.bgz.gz"
    bgz.write_bytes(bamgen.bgzf(data[:1 << 20], block=0x4000, level=1)[:-28] + bamgen.bgzf(data[1 << 20:], level=1))  # small and full-size blocks
    bgz_env = dict(ENV, FQGPU_HOST_THREADS="3", TSAN_OPTIONS="halt_on_error=1")
    for piece in ("4096", "100000", "50000000"):
        for mode in ("0", "1", "2"):
            for extra in ({}, {"FQGPU_NO_PARALLEL_INFLATE": "1"}):
                p = subprocess.run([exe, str(bgz), piece, mode], env=dict(bgz_env, **extra), capture_output=True, timeout=300)
                assert p.returncode == 0 and p.stdout.strip() == want, (piece, mode, extra, p.stdout, p.stderr.decode()[-1500:])
    broken = bytearray(bgz.read_bytes())
    broken[len(broken) // 2] ^= 0x55  # a flipped byte in the middle of some block: inflate or CRC-32 must notice
    bad = tmpdir / "broken.bgz.gz"
    bad.write_bytes(bytes(broken))
    p = subprocess.run([exe, str(bad), "100000", "0"], env=bgz_env, capture_output=True, timeout=300)
    assert p.returncode != 0 and (b"BGZF" in p.stderr), p.stderr.decode()[-800:]
    # any other gzip file beyond 1 MiB is inflated by chunks whose first blocks are searched for (fq_pgzip.h); small chunks
    # here, so that this file is dozens of them, and a file of several members with bytes behind the last one
    big = tmpdir / "big.txt.gz"
    big.write_bytes(gzip.compress(data, 6) + gzip.compress(data[:300000], 1) + b"trailing bytes")
    want_big = ("%d %d" % (len(data) + 300000, fnv(data + data[:300000]))).encode()
    assert big.stat().st_size > (1 << 20)
    for piece in ("100000", "50000000"):
        for mode in ("0", "2"):
            for extra in ({"FQGPU_PGZIP_CHUNK": "30000"}, {}, {"FQGPU_NO_PARALLEL_INFLATE": "1"}):
                p = subprocess.run([exe, str(big), piece, mode], env=dict(bgz_env, FQGPU_PGZIP_DEBUG="1", **extra), capture_output=True, timeout=300)
                assert p.returncode == 0 and p.stdout.strip() == want_big, (piece, mode, extra, p.stdout, p.stderr.decode()[-1500:])
                assert (b"inflated by chunks" in p.stderr) == ("FQGPU_NO_PARALLEL_INFLATE" not in extra), p.stderr.decode()[-800:]
                if "FQGPU_PGZIP_CHUNK" in extra:
                    assert b"one zlib stream" not in p.stderr and b" 2 members" in p.stderr, p.stderr.decode()[-800:]
    # a file that inflates to far more than 24 times its size: its slots are sized by the file (piece_for_file), so it
    # comes in several pieces where a piece of 50 MB was asked for - and whole when asked for whole
    rep_data = b"ACGTACGTAC\n" * 400000
    rep = tmpdir / "rep.txt.gz"
    rep.write_bytes(gzip.compress(rep_data, 6))
    assert rep.stat().st_size * 24 < len(rep_data) // 2
    want_rep = ("%d %d" % (len(rep_data), fnv(rep_data))).encode()
    for mode in ("0", "1", "2"):
        for extra in ({}, {"FQGPU_NO_PARALLEL_INFLATE": "1"}):
            p = subprocess.run([exe, str(rep), "50000000", mode], env=dict(bgz_env, **extra), capture_output=True, timeout=300)
            assert p.returncode == 0 and p.stdout.strip() == want_rep, (mode, extra, p.stdout, p.stderr.decode()[-1500:])
    for path, wanted in ((plain, want), (gz, want), (empty, ("0 %d" % fnv(b"")).encode())):
        for piece in ("4096", "100000", "50000000"):
            for mode in ("0", "1", "2"):
                p = subprocess.run([exe, str(path), piece, mode], env=dict(ENV, FQGPU_HOST_THREADS="3", TSAN_OPTIONS="halt_on_error=1"),
                                   capture_output=True, timeout=300)
                assert p.returncode == 0 and p.stdout.strip() == wanted, (path, piece, mode, p.stdout, p.stderr.decode()[-1500:])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_record_aligned_piece_cutter_runs_clean(tmpdir, san):
    """fq_multi.h (FQGPU_DEVICES): pieces cut on the host by counting lines must tile the file, start at records
    4*first_record and - all but the last - hold whole records; several consumers, plain / gz / truncated / empty."""
    flags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g", "-O1"]
    exe = str(tmpdir / ("pieces_" + san.split(",")[0]))
    subprocess.run(["g++", "-std=c++17", "-pthread"] + flags + ["-o", exe, os.path.join(CXX, "pieces_check.cpp"), "-lz"], check=True)
    rng = np.random.default_rng(5)
    recs = []
    for i in range(4000):
        n = int(rng.integers(1, 300))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * n + b"\n")
    data = b"".join(recs)
    files = {"full.fq": data, "full.fq.gz": gzip.compress(data, 1), "cut.fq": data[:-57], "nonl.fq": data[:-1],
             "empty.fq": b"", "one.fq": recs[0], "blank.fq": b"\n" * 1001}
    # (a file that inflates to far more than 24 times its size: the slots are sized by the file - piece_for_file -, so
    # it comes in several pieces whatever piece was asked for)
    rep = b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n" * 120000
    files["rep.fq.gz"] = gzip.compress(rep, 6)
    assert len(files["rep.fq.gz"]) * 24 < len(rep) // 2
    inflated = {"full.fq.gz": len(data), "rep.fq.gz": len(rep)}
    for name, content in files.items():
        (tmpdir / name).write_bytes(content)
        size = inflated.get(name, len(content))
        # (pieces of 512 bytes only for the small files: a thousand hand-overs per run under the sanitizers' run times are
        # what made these two tests half of the CPU suite's time)
        for piece in (("100000", "50000000") if name == "rep.fq.gz" else
                      ("512", "4096", "100000", "50000000") if len(content) < 100000 else ("4096", "100000", "50000000")):
            for consumers in ("1", "3"):
                # (a .gz once more through the many-core gzip reader, fq_pgzip.h, in chunks of 20 kB: files this small are
                # one zlib thread's otherwise)
                for extra in ({}, {"FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "20000", "FQGPU_PGZIP_DEBUG": "1"}) if name.endswith(".gz") else ({},):
                    p = subprocess.run([exe, str(tmpdir / name), piece, consumers],
                                       env=dict(ENV, FQGPU_HOST_THREADS="3", TSAN_OPTIONS="halt_on_error=1", **extra), capture_output=True, timeout=300)
                    out = p.stdout.decode().split()
                    assert p.returncode == 0 and out[-1] == "ok" and int(out[1]) == size, (name, piece, consumers, p.stdout, p.stderr.decode()[-1500:])
                    assert (b"inflated by chunks" in p.stderr) == bool(extra), p.stderr.decode()[-500:]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_record_block_cutter_runs_clean(tmpdir, san):
    """fq_blocks.h (fastq_pre_barcodes with FQGPU_DEVICES): blocks of exactly B records cut on the host by counting lines
    must tile the file and start at record k*B; several consumers, plain / gz / truncated / empty, blocks smaller and
    larger than the first bytes the cutter peeks at, files that end exactly at a block boundary."""
    flags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g", "-O1"]
    exe = str(tmpdir / ("blocks_" + san.split(",")[0]))
    subprocess.run(["g++", "-std=c++17", "-pthread"] + flags + ["-o", exe, os.path.join(CXX, "blocks_check.cpp"), "-lz"], check=True)
    rng = np.random.default_rng(6)
    recs = []
    for i in range(6000):
        n = int(rng.integers(1, 300))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * n + b"\n")
    data = b"".join(recs)
    files = {"full.fq": data, "full.fq.gz": gzip.compress(data, 1), "cut.fq": data[:-57], "nonl.fq": data[:-1],
             "empty.fq": b"", "one.fq": recs[0], "blank.fq": b"\n" * 1001, "exact.fq": b"".join(recs[:3000]),
             "small.fq": b"".join(recs[:50])}
    for name, content in files.items():
        (tmpdir / name).write_bytes(content)
        size = len(data if name.endswith(".gz") else content)
        for per_block in ("1", "7", "1000", "3000", "100000"):
            if per_block in ("1", "7") and size > 100000:
                continue
            for consumers in ("1", "3"):
                for extra in ({}, {"FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "20000", "FQGPU_PGZIP_DEBUG": "1"}) if name.endswith(".gz") else ({},):
                    p = subprocess.run([exe, str(tmpdir / name), per_block, consumers],
                                       env=dict(ENV, FQGPU_HOST_THREADS="3", TSAN_OPTIONS="halt_on_error=1", **extra), capture_output=True, timeout=300)
                    out = p.stdout.decode().split()
                    assert p.returncode == 0 and out[-1] == "ok" and int(out[1]) == size, (name, per_block, consumers, p.stdout, p.stderr.decode()[-1500:])
                    assert (b"inflated by chunks" in p.stderr) == bool(extra), p.stderr.decode()[-500:]
