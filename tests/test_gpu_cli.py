"""End-to-end parity of the drop-in program bin/fastq_info (C++ host + libfqgpu.so) with the
reference program: every golden invocation (tests/golden/fastq_info.json, captured from the
reference binary) must give the same exit status, stdout and stderr; seeded mutated inputs are
checked against the oracle in all four modes."""
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import loader as orc
from tests import fuzz
from tests.util import GOLD, REPO, SideBySide, load_fastq_info_golden, strip_progress, thinned

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "bin", "fastq_info")
GOLDEN = load_fastq_info_golden()


def run_cli(args, cwd, env_extra=None):
    env = dict(os.environ)
    if env_extra:
        env.update(env_extra)
    p = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, timeout=300, env=env)
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


def test_binary_exists():
    assert os.path.exists(BIN), "run __graft_entry__.build() first"


POOL = 12  # programs started side by side: a process start + HIP initialisation is 0.34 s alone, and the driver takes
# about 15 of them a second however many wait (tools/golden_concurrency.py) - one at a time the suite does not fit its slot
CHUNKED = {"FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "4096", "FQGPU_HOST_THREADS": "3"}


def test_all_golden_invocations():
    """Every golden invocation, once.  The goldens' inputs are .fastq.gz files: every second case has them read by the
    many-core gzip reader (host/fq_pgzip.h) in chunks of 4 KiB, the others by the one zlib thread that files this small
    get otherwise (a checksum of the arguments decides - both paths see every kind of invocation)."""
    import zlib

    def one(case):
        chunked = zlib.crc32(" ".join(case["args"]).encode()) & 1
        rc, out, err = run_cli(case["args"], GOLD, CHUNKED if chunked else None)
        ok = (rc == case["exit"] and out == case["stdout"]
              and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], chunked, rc, case["exit"], err[-400:], case["stderr"][-400:])

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, GOLDEN) if b]
    assert not bad, f"{len(bad)} of {len(GOLDEN)} differ; first: {bad[:3]}"


def test_gzip_files_of_many_chunks_in_every_mode():
    """paired files of 30 000 reads as .gz (levels 1 and 9, one of them of three members): dozens of chunks of 64 KiB per
    file, pieces of 1 MiB on the GPU side; all modes against the oracle, and the timing line must say that the chunks
    were found and joined"""
    import gzip

    with tempfile.TemporaryDirectory() as tmp:
        a = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=1)
        b = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=2)
        dup = a + b"\n".join(a.split(b"\n")[4 * 123:4 * 123 + 4]) + b"\n"
        files = {"a.fastq.gz": a, "b.fastq.gz": b, "d.fastq.gz": dup}
        third = len(b) // 3
        packed = {"a.fastq.gz": gzip.compress(a, 9), "d.fastq.gz": gzip.compress(dup, 1),
                  "b.fastq.gz": gzip.compress(b[:third], 6) + gzip.compress(b[third:2 * third], 1) + gzip.compress(b[2 * third:], 6)}
        for name, blob in packed.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(blob)
        env = {"FQGPU_CHUNK_MB": "1", "FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "65536", "FQGPU_HOST_THREADS": "4"}
        compare_all([(tmp, args, files, env) for args in (["-r", "a.fastq.gz"], ["a.fastq.gz"], ["a.fastq.gz", "b.fastq.gz"],
                                                          ["d.fastq.gz"], ["-s", "a.fastq.gz", "b.fastq.gz"])])
        rc, out, err = run_cli(["-r", "b.fastq.gz"], tmp, dict(env, FQGPU_TIMING="1"))
        line = [ln for ln in err.splitlines() if "inflated by chunks" in ln]
        assert rc == 0 and line and " 3 members" in line[0] and "one zlib stream" not in line[0], err[-800:]
        joined = int(line[0].split(" chunks joined")[0].split()[-1])
        assert joined >= 20, line[0]


def oracle_run(args, files):
    flags, pos = orc.parse_args(args)
    if len(pos) == 1:
        return orc.fastq_info(files[pos[0]], pos[0], flags=flags)
    if pos[1].startswith("pe"):
        return orc.fastq_info(files[pos[0]], pos[0], None, pos[1], orc.ARG2_PE, flags)
    return orc.fastq_info(files[pos[0]], pos[0], files[pos[1]], pos[1], orc.ARG2_FILE, flags)


def compare_with_oracle(tmp, args, files, env=None):
    rc, out, err = run_cli(args, tmp, env)
    want = oracle_run(args, files)
    ctx = (args, err[-500:], want["stderr"][-500:])
    assert rc == want["exit"], ctx
    assert out == want["stdout"], ctx
    assert strip_progress(err) == strip_progress(want["stderr"]), ctx


def compare_all(jobs):
    """jobs = (cwd, args, files, env): the programs run side by side, every one compared with the oracle"""
    def one(job):
        cwd, args, files, env = job
        rc, out, err = run_cli(args, cwd, env)
        want = oracle_run(args, files)
        if rc == want["exit"] and out == want["stdout"] and strip_progress(err) == strip_progress(want["stderr"]):
            return None
        return (os.path.basename(cwd), args, env, rc, want["exit"], err[-500:], want["stderr"][-500:])

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, jobs) if b]
    assert not bad, f"{len(bad)} of {len(jobs)} differ; first: {bad[:2]}"


def put(tmp, sub, files):
    """a directory of its own for a set of files (the runs that read them start later, side by side)"""
    d = os.path.join(tmp, sub)
    os.makedirs(d, exist_ok=True)
    for name, img in files.items():
        with open(os.path.join(d, name), "wb") as f:
            f.write(img)
    return d


@pytest.mark.parametrize("kind", fuzz.MUTATIONS)
def test_mutated_single_files(kind):
    rng = np.random.default_rng(abs(hash("cli" + kind)) % 100000)
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(4):
            style = ["casava", "slash", "int", "nosuffix"][trial % 4]
            img = fuzz.make_fastq(rng, int(rng.integers(1, 300)), 1, 100, style, hdr2_names=bool(trial & 1),
                                  crlf=(trial == 3), rna=(trial == 2))
            img = fuzz.mutate(rng, img, kind)
            d = put(tmp, "t%d" % trial, {"f.fastq": img})
            jobs += [(d, args, {"f.fastq": img}, None) for args in (["-r", "f.fastq"], ["f.fastq"], ["f.fastq", "pe"])]
        compare_all(jobs)


def test_duplicates_and_pairs():
    rng = np.random.default_rng(5)
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(10):
            style = ["casava", "slash"][trial % 2]
            n = int(rng.integers(2, 400))
            a = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 60, style, mate=1)
            b = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 60, style, mate=2)
            la, lb = a.split(b"\n"), b.split(b"\n")
            if trial % 5 == 1:  # duplicate a record of file 1 (twice, to test "earliest repeat")
                k = int(rng.integers(0, n))
                la = la[:-1] + la[4 * k:4 * k + 4] + la[4 * k:4 * k + 4] + [b""]
            if trial % 5 == 2:  # drop a record from file 2
                k = int(rng.integers(0, n))
                lb = lb[:4 * k] + lb[4 * k + 4:]
            if trial % 5 == 3:  # shuffle file 2 (still paired)
                recs = [lb[4 * i:4 * i + 4] for i in range(n)]
                order = rng.permutation(n)
                lb = [x for i in order for x in recs[i]] + [b""]
            if trial % 5 == 4:  # duplicate in file 2
                k = int(rng.integers(0, n))
                lb = lb[:-1] + lb[4 * k:4 * k + 4] + [b""]
            a, b = b"\n".join(la), b"\n".join(lb)
            files = {"a.fastq": a, "b.fastq": b}
            d = put(tmp, "t%d" % trial, files)
            jobs += [(d, args, files, None) for args in (["a.fastq"], ["a.fastq", "b.fastq"], ["b.fastq", "a.fastq"],
                                                         ["-r", "-s", "a.fastq", "b.fastq"], ["-s", "a.fastq", "b.fastq"])]
        compare_all(jobs)


def test_small_pieces_exercise_the_carry():
    """1 MiB pieces: records straddle piece boundaries, the index spans many segments."""
    rng = np.random.default_rng(17)
    with tempfile.TemporaryDirectory() as tmp:
        a = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=1)
        b = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=2)
        dup = a + b"\n".join(a.split(b"\n")[4 * 123:4 * 123 + 4]) + b"\n"
        files = {"a.fastq": a, "b.fastq": b, "d.fastq": dup}
        for name, img in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(img)
        env = {"FQGPU_CHUNK_MB": "1"}
        compare_all([(tmp, args, files, env) for args in (["-r", "a.fastq"], ["a.fastq"], ["a.fastq", "b.fastq"], ["d.fastq"])])


# ---- FQGPU_DEVICES: the -r pass over several contexts (host/fq_multi.h) ---------------------------------------------
# (the test box has one GPU: "0,0,0" opens three contexts on it - the same threads, pieces, result order and
# accumulator merge as three GPUs)
MULTI = {"FQGPU_DEVICES": "0,0,0"}


def test_several_devices_golden_dash_r_invocations():
    cases = [c for c in GOLDEN if "-r" in c["args"] and len([a for a in c["args"] if not a.startswith("-")]) == 1]
    assert len(cases) > 20
    cases = thinned(cases)

    def one(case):
        rc, out, err = run_cli(case["args"], GOLD, MULTI)
        ok = (rc == case["exit"] and out == case["stdout"]
              and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], rc, case["exit"], err[-400:], case["stderr"][-400:])

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, cases) if b]
    assert not bad, f"{len(bad)} of {len(cases)} differ; first: {bad[:3]}"


@pytest.mark.parametrize("kind", ["clean"] + fuzz.MUTATIONS)
def test_several_devices_many_pieces(kind):
    """1 MiB pieces of a ~9 MB file: ~9 pieces over three contexts; a finding anywhere (or none: merged statistics)
    must read exactly as the serial loop's."""
    rng = np.random.default_rng(abs(hash("multi" + kind)) % 100000)
    env = dict(MULTI, FQGPU_CHUNK_MB="1")
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(2):
            img = fuzz.make_fastq(rng, 30000, 20 if trial else 100, 250 if trial else 101, ["casava", "slash"][trial])
            if kind != "clean":
                img = fuzz.mutate(rng, img, kind)
            d = put(tmp, "t%d" % trial, {"f.fastq": img})
            jobs.append((d, ["-r", "f.fastq"], {"f.fastq": img}, env))
            if trial == 0:
                import gzip
                put(tmp, "t0", {"g.fastq.gz": gzip.compress(img, 1)})
                jobs.append((d, ["-r", "g.fastq.gz"], {"g.fastq.gz": img}, env))
        compare_all(jobs)


REF_INFO = os.path.join(REPO, "oracle", "_ref", "fastq_info")


def _overlong():
    from tests.test_oracle_vs_ref_fuzz import overlong_images

    return overlong_images()


LONG_HOW = {"plain_file": None, "gz_file": None, "gz_file_by_chunks": dict(CHUNKED, FQGPU_PGZIP_CHUNK="30000"),
            "small_pieces": {"FQGPU_CHUNK_MB": "1"}, "several_devices": {"FQGPU_DEVICES": "0,0", "FQGPU_CHUNK_MB": "1"}}
LONG_ARGS = (["-r", "F"], ["F"], ["F", "pe"], ["-r", "-s", "F", "F"], ["F", "F"])
LONG_ROOT = tempfile.TemporaryDirectory()


def _long_line_run(key):
    """one invocation on a file of its own: (what the program said, what the oracle says, what the reference binary said)"""
    import gzip

    which, how, a = key
    img = _overlong()[which]
    name = "f.fastq.gz" if how.startswith("gz_file") else "f.fastq"
    d = tempfile.mkdtemp(dir=LONG_ROOT.name)
    with open(os.path.join(d, name), "wb") as f:
        f.write(gzip.compress(img, 1) if how.startswith("gz_file") else img)
    args = [name if x == "F" else x for x in LONG_ARGS[a]]
    got = run_cli(args, d, LONG_HOW[how])
    want = oracle_run(args, {name: img})
    ref = None
    if os.path.exists(REF_INFO):
        p = subprocess.run([REF_INFO] + args, cwd=d, capture_output=True, timeout=300)
        ref = (p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1"))
    return args, got, want, ref


# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
LONG_RUNS = SideBySide(_long_line_run, [(w, h, a) for w in sorted(_overlong()) for h in LONG_HOW for a in range(len(LONG_ARGS))],
                       select=lambda ks: [k for k in ks if k[1] == list(LONG_HOW)[0]] +
                       [k for k in ks if k[1] != list(LONG_HOW)[0] and (k[0], k[1]) in set(thinned({(w, h) for w, h, _ in ks if h != list(LONG_HOW)[0]}, key="-".join))])


@pytest.mark.parametrize("how", list(LONG_HOW))
@pytest.mark.parametrize("which", sorted(_overlong()))
def test_lines_beyond_the_gzgets_buffers(which, how):
    """A line beyond the reference's gzgets buffers (src/fastq.c:249-253: 1000 bytes for the header lines, 2 500 000 for
    sequence and quality) is read in pieces there, which puts every later line out of step - deterministic behaviour,
    and bin/fastq_info reproduces it: the input is cut the way gzgets cuts it (host/fq_reframe.h) - while it is read
    when it is inflated anyway, and in a second run of the program when the GPU finds such a line in a plain file
    (host/fq_respawn.h: nothing is printed twice) - and the GPU sees the pieces as lines.  Exit status, stdout and
    stderr of the oracle (pinned on the reference binary for these very images, tests/test_oracle_vs_ref_fuzz.py) and,
    byte for byte with the progress ticker, of the reference binary itself."""
    for a in range(len(LONG_ARGS)):
        args, (rc, out, err), want, ref = LONG_RUNS.get((which, how, a))
        ctx = (args, err[-500:], want["stderr"][-500:])
        assert rc == want["exit"], ctx
        assert out == want["stdout"], ctx
        assert strip_progress(err) == strip_progress(want["stderr"]), ctx
        if ref is not None:
            assert (rc, out, err) == ref, ctx


@pytest.mark.skipif(not os.path.exists(REF_INFO), reason="oracle/_ref not built")
@pytest.mark.parametrize("args", [["-r", "f.fastq"], ["f.fastq"]], ids=["r", "index"])
def test_a_long_line_late_in_a_file_prints_nothing_twice(args):
    """250 000 reads (the progress ticker has written twice), then a header beyond the limit: the run that finds it has
    printed the version line, the format line and the ticker; the run that takes over prints the rest.  stderr must be
    the reference's, byte for byte."""
    rng = np.random.default_rng(8)
    ok = fuzz.make_fastq(rng, 250_000, 20, 40, "casava")
    img = ok + b"@" + b"h" * 1500 + b" 1:N:0:A\nACGT\n+\nIIII\n" + fuzz.make_fastq(rng, 10, 20, 40, "casava")
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "f.fastq"), "wb") as f:
            f.write(img)
        p = subprocess.run([REF_INFO] + args, cwd=tmp, capture_output=True, timeout=600)
        for env in (None, {"FQGPU_CHUNK_MB": "4"}):
            rc, out, err = run_cli(args, tmp, env)
            assert (rc, out) == (p.returncode, p.stdout.decode("latin-1"))
            assert err == p.stderr.decode("latin-1"), (err[-300:], p.stderr[-300:])


def test_bgzipped_input_and_json_metrics():
    """A bgzip'd FASTQ file (BGZF blocks, inflated on many threads by host/fq_input.h) gives what the same bytes give as a
    plain file and as an ordinary .gz - in every mode - and FQGPU_JSON_METRICS writes the machine-readable twin of the
    summary without touching stdout / stderr."""
    import gzip
    import json

    from tests import bamgen

    rng = np.random.default_rng(21)
    img = fuzz.make_fastq(rng, 60_000, 30, 120, "casava")
    bad = fuzz.mutate(rng, img, "flip_seq")
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in (("a", img), ("b", bad)):
            with open(os.path.join(tmp, name + ".fastq"), "wb") as f:
                f.write(data)
            with open(os.path.join(tmp, name + ".bgz.fastq.gz"), "wb") as f:
                f.write(bamgen.bgzf(data, level=1))
            with open(os.path.join(tmp, name + ".z.fastq.gz"), "wb") as f:
                f.write(gzip.compress(data, 1))
        runs = [(stem, mode, ext) for stem in ("a", "b") for mode in (["-r"], [], ["pe"]) for ext in (".fastq", ".bgz.fastq.gz", ".z.fastq.gz")]

        def one(run):
            stem, mode, ext = run
            args = [a for a in mode if a != "pe"] + [stem + ext] + (["pe"] if "pe" in mode else [])
            rc, out, err = run_cli(args, tmp, {"FQGPU_CHUNK_MB": "1"})
            return (rc, out, strip_progress(err).replace(stem + ext, "F"))

        with ThreadPoolExecutor(POOL) as ex:
            got = list(ex.map(one, runs))
        for k in range(0, len(runs), 3):  # the three containers of one (file, mode)
            assert got[k] == got[k + 1] == got[k + 2], (runs[k], got[k][2][-300:], got[k + 1][2][-300:], got[k + 2][2][-300:])
        jm = os.path.join(tmp, "m.json")
        rc, out, err = run_cli(["-r", "a.bgz.fastq.gz"], tmp, {"FQGPU_JSON_METRICS": jm})
        rc0, out0, err0 = run_cli(["-r", "a.bgz.fastq.gz"], tmp)
        assert (rc, out, err) == (rc0, out0, err0) and rc == 0
        m = json.load(open(jm))
        assert m["program"] == "fastq_info" and m["reads"] == 60_000 and m["input_bytes"] == len(img)
        assert m["seconds"] > 0 and m["Mreads_per_s"] > 0 and m["min_read_length"] == 30 and m["max_read_length"] == 120


# ---- FQGPU_DEVICES in the index modes: names across contexts (host/fq_names_multi.h) ----------------------------------
def goes_to_several_devices(case):
    """The golden invocations that test names are run once more on two other paths: every second one (a checksum of its
    arguments decides) over several devices, here; the others through the streaming pass with header capture,
    tests/test_gpu_name_capture.py.  (A program start is what the suite's time is made of: tools/golden_concurrency.py.)"""
    import zlib

    return (zlib.crc32(("paths " + " ".join(case["args"])).encode()) >> 3) & 1 == 1


def test_several_devices_golden_index_and_pairing_invocations():
    """golden invocations that test names (no -r, not interleaved "pe"): records spread over three contexts, names
    tested across them by the fingerprint exchange - same exit status, stdout and stderr as the reference binary"""
    cases = [c for c in GOLDEN if "-r" not in c["args"] and "pe" not in c["args"] and goes_to_several_devices(c)]
    assert len(cases) > 75
    cases = thinned(cases)

    def one(case):
        rc, out, err = run_cli(case["args"], GOLD, MULTI)
        ok = (rc == case["exit"] and out == case["stdout"]
              and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], rc, case["exit"], err[-400:], case["stderr"][-400:])

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, cases) if b]
    assert not bad, f"{len(bad)} of {len(cases)} differ; first: {bad[:3]}"


def test_several_devices_duplicates_and_pairs_in_many_pieces():
    """1 MiB pieces over three contexts: a name repeated pieces apart, a missing mate, mates in another order, a name asked
    for twice - what the serial loops print"""
    rng = np.random.default_rng(23)
    env = dict(MULTI, FQGPU_CHUNK_MB="1")
    with tempfile.TemporaryDirectory() as tmp:
        n = 30000
        a = fuzz.make_fastq(np.random.default_rng(3), n, 50, 150, "casava", mate=1)
        b = fuzz.make_fastq(np.random.default_rng(3), n, 50, 150, "casava", mate=2)
        la, lb = a.split(b"\n"), b.split(b"\n")
        recs_b = [lb[4 * i:4 * i + 4] for i in range(n)]
        dup = b"\n".join(la[:4 * 20000] + la[4 * 77:4 * 78] + la[4 * 20000:])              # record 77 again as record 20000
        missing = b"\n".join(lb[:4 * 12345] + lb[4 * 12346:])                               # a mate less in file 2
        shuffled = b"\n".join([x for i in rng.permutation(n) for x in recs_b[i]] + [b""])   # still paired
        twice = b"\n".join(lb[:4 * 25000] + lb[4 * 100:4 * 101] + lb[4 * 25000:])           # a name asked for twice
        bad_base = bytearray(a)
        bad_base[len(a) // 2 + a[len(a) // 2:].index(b"\n+\n") - 3] = ord("X")              # a validation finding mid-file
        files = {"a.fastq": a, "b.fastq": b, "d.fastq": dup, "m.fastq": missing, "s.fastq": shuffled, "t.fastq": twice,
                 "x.fastq": bytes(bad_base)}
        for name, img in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(img)
        compare_all([(tmp, args, files, env) for args in (
            ["a.fastq"], ["d.fastq"], ["x.fastq"], ["a.fastq", "b.fastq"], ["a.fastq", "m.fastq"], ["m.fastq", "a.fastq"],
            ["a.fastq", "s.fastq"], ["a.fastq", "t.fastq"], ["x.fastq", "b.fastq"], ["a.fastq", "x.fastq"], ["d.fastq", "b.fastq"])])
        # Reads without a mate on BOTH sides, with fingerprints of 16 bits and no check bits (FQGPU_FP_WEAK_BITS, a test
        # hook): a holder and an asker with DIFFERENT names under one fingerprint are common then, and what says "the
        # same name" is the names themselves, which travel beside the pairs (host/fq_names_multi.h)
        recs_a = [la[4 * i:4 * i + 4] for i in range(n)]
        files["a7.fastq"] = b"\n".join([x for i in range(n) if i % 7 for x in recs_a[i]] + [b""])
        files["b11.fastq"] = b"\n".join([x for i in range(n) if i % 11 for x in recs_b[i]] + [b""])
        put(tmp, ".", {k: files[k] for k in ("a7.fastq", "b11.fastq")})
        weak = dict(env, FQGPU_FP_WEAK_BITS="16")
        compare_all([(tmp, args, files, e) for e in (env, weak) for args in (["a7.fastq", "b11.fastq"], ["b11.fastq", "a7.fastq"])])


@pytest.mark.parametrize("kind", fuzz.MUTATIONS)
def test_several_devices_mutated_files_in_index_mode(kind):
    rng = np.random.default_rng(abs(hash("multi-index" + kind)) % 100000)
    env = dict(MULTI, FQGPU_CHUNK_MB="1")
    with tempfile.TemporaryDirectory() as tmp:
        img = fuzz.make_fastq(rng, 30000, 20, 120, "casava")
        img = fuzz.mutate(rng, img, kind)
        mate = fuzz.make_fastq(np.random.default_rng(5), 2000, 20, 120, "casava", mate=2)
        files = {"f.fastq": img, "g.fastq": mate}
        for name, data in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(data)
        compare_all([(tmp, ["f.fastq"], files, env), (tmp, ["g.fastq", "f.fastq"], files, env)])


@pytest.mark.parametrize("n_reads", [300, 30000], ids=["one_piece", "many_pieces"])
def test_several_devices_a_nul_byte_at_a_record_start_ends_the_file(n_reads):
    """the reference reads C strings: a record whose first byte is NUL ends the file there (src/fastq.c:250).  Over several
    devices the later pieces have been looked at by then - the file is passed over again up to that byte; what is printed
    once by the serial loop is printed once (found by tools/fuzz_campaign.py, seeds 1000 and 1098)"""
    a = fuzz.make_fastq(np.random.default_rng(3), n_reads, 20, 120, "int", mate=1)
    b = fuzz.make_fastq(np.random.default_rng(3), n_reads, 20, 120, "int", mate=2)
    cut = n_reads * 2 // 3

    def nul_at(img, rec):
        lines = img.split(b"\n")
        lines[4 * rec] = b"\0" + lines[4 * rec][1:]
        return b"\n".join(lines)

    files = {"a.fastq": a, "b.fastq": b, "an.fastq": nul_at(a, cut), "bn.fastq": nul_at(b, cut), "a0.fastq": nul_at(a, 0)}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(data)
        compare_all([(tmp, args, files, env)
                     for env in (dict(MULTI, FQGPU_CHUNK_MB="1"), dict(MULTI, FQGPU_CHUNK_MB="1", FQGPU_STREAM_MIN="256"), {"FQGPU_CHUNK_MB": "1"})
                     for args in (["-r", "an.fastq"], ["an.fastq"], ["an.fastq", "pe"], ["a.fastq", "bn.fastq"], ["an.fastq", "b.fastq"],
                                  ["an.fastq", "bn.fastq"], ["bn.fastq", "a.fastq"], ["-r", "a0.fastq"], ["a0.fastq"], ["a.fastq", "a0.fastq"])])


def test_paired_sorted_mode_reads_on_behind_a_nul_line():
    """-r -s: a line that starts with a NUL byte makes fastq_read_entry return "no entry", which ends the loop - and the two
    reads after the loop (src/fastq_info.c:142-149) carry on BEHIND that line: a whole record there is "Premature end of
    file2" / "file1", three lines are a truncated file, nothing is a clean end (tools/fuzz_campaign.py, seed 30415)"""
    n = 60
    a = fuzz.make_fastq(np.random.default_rng(8), n, 10, 60, "nosuffix", mate=1)
    b = fuzz.make_fastq(np.random.default_rng(8), n, 10, 60, "nosuffix", mate=2)

    def lines_of(img):
        return img.split(b"\n")[:-1]

    def with_nul(img, rec, keep_after):
        """record `rec` starts with a NUL byte; only keep_after lines of the file follow that line (None: all)"""
        ls = lines_of(img)
        ls[4 * rec] = b"\0" + ls[4 * rec]
        if keep_after is not None:
            ls = ls[:4 * rec + 1 + keep_after]
        return b"\n".join(ls) + b"\n"

    shapes = {"a_mid": (with_nul(a, 20, None), b), "a_mid_b_short": (with_nul(a, 20, None), b"\n".join(lines_of(b)[:4 * 20]) + b"\n"),
              "a_last_3_lines": (with_nul(a, 59, 3), b), "a_then_4_lines": (with_nul(a, 30, 4), b), "a_then_2_lines": (with_nul(a, 30, 2), b),
              "a_then_nothing": (with_nul(a, 30, 0), b), "a_then_nothing_b_same": (with_nul(a, 30, 0), b"\n".join(lines_of(b)[:4 * 30]) + b"\n"),
              "a_first": (with_nul(a, 0, None), b), "a_two_nul_lines": (with_nul(with_nul(a, 31, None), 30, 1), b),
              "b_mid": (a, with_nul(b, 20, None)), "b_mid_a_short": (b"\n".join(lines_of(a)[:4 * 21]) + b"\n", with_nul(b, 20, None)),
              "b_then_nothing": (a, with_nul(b, 20, 0)), "b_then_nothing_a_same": (b"\n".join(lines_of(a)[:4 * 21]) + b"\n", with_nul(b, 20, 0)),
              "b_then_3_lines": (b"\n".join(lines_of(a)[:4 * 21]) + b"\n", with_nul(b, 20, 3)), "b_first": (a, with_nul(b, 0, None)),
              "both_same_record": (with_nul(a, 25, None), with_nul(b, 25, None)), "a_before_b": (with_nul(a, 24, 0), with_nul(b, 25, None))}
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, (fa, fb) in shapes.items():
            files = {"a.fastq": fa, "b.fastq": fb}
            d = put(tmp, tag, files)
            jobs += [(d, args, files, env) for env in ({}, {"FQGPU_STREAM_MIN": "256"})
                     for args in (["-r", "-s", "a.fastq", "b.fastq"], ["-r", "-s", "b.fastq", "a.fastq"], ["-s", "a.fastq", "b.fastq"])]
        compare_all(jobs)
