"""GPU parity tests for framing + validation (fqg_validate), through the C-ABI.

Oracle: oracle/fq_oracle.c running `fastq_info -r` semantics (validate_single_fastq_file,
reference src/fastq_info.c:155-176: is_pe=TRUE, every record read then validated).  The HIP
path must report the same first outcome (code, record, aux) and, when the file is clean, the
same statistics.  Integer / byte work: the bar is bit-exact.
"""
import os

import numpy as np
import pytest

from oracle import loader as orc
from tests import fuzz
from tests.util import GOLD, read_image

pytestmark = pytest.mark.gpu

fq = pytest.importorskip("fastq_utils_amd")


@pytest.fixture(scope="module", params=["default", "stream"])
def ctx(request):
    """Every test runs twice: with the library's defaults (images below 1 MiB take the two-pass
    framing path) and with the single-pass streaming path forced on every image of >= 256 bytes."""
    old = os.environ.get("FQGPU_STREAM_MIN")
    if request.param == "stream":
        os.environ["FQGPU_STREAM_MIN"] = "256"
    else:
        os.environ.pop("FQGPU_STREAM_MIN", None)
    c = fq.Context(0)  # the threshold is read when the context is opened
    if old is None:
        os.environ.pop("FQGPU_STREAM_MIN", None)
    else:
        os.environ["FQGPU_STREAM_MIN"] = old
    yield c
    c.close()


def expected_from_oracle(image, flags=orc.FLAG_R):
    return orc.fastq_info(image, "x.fastq", flags=flags | orc.FLAG_Q)


def check_image(ctx, image, force_exact=False):
    """Run both sides on one image and compare.  Returns the product result."""
    want = expected_from_oracle(image)
    st = fq.abi.probe_first_record(image, True)
    acc = ctx.accumulator()
    try:
        flags = fq.abi.VALIDATE_FORCE_EXACT if force_exact else 0
        got = ctx.validate(image, acc, st, final=True, flags=flags)
        first = want["first"]
        ctxmsg = f"oracle={first} exit={want['exit']} got={got} stderr={want['stderr'][-300:]!r}"
        if first["code"] != 0:
            assert got["code"] == first["code"], ctxmsg
            assert got["record"] == first["record"], ctxmsg
            if first["code"] in (6, 8, 11, 12):
                assert got["aux0"] == first["aux0"], ctxmsg
            if first["code"] in (11, 12):
                assert got["aux1"] == first["aux1"], ctxmsg
        else:
            assert got["code"] == 0, ctxmsg
            if "No reads found" in want["stderr"]:
                assert got["n_records"] == 0, ctxmsg
            else:
                assert want["exit"] == 0, ctxmsg
                s = want["summary"]
                stats = acc.read()
                assert got["n_records"] == s["num_reads"], ctxmsg
                assert stats["num_rds"] == s["num_reads"], ctxmsg
                assert (stats["min_rl"], stats["max_rl"]) == (s["min_rl"], s["max_rl"]), ctxmsg
                assert (stats["min_qual"], stats["max_qual"]) == (s["min_qual"], s["max_qual"]), ctxmsg
                assert acc.median() == s["median_rl"], ctxmsg
        return got
    finally:
        acc.close()


FIXTURES = sorted(f for f in os.listdir(os.path.join(GOLD, "data")))


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("force_exact", [True, False])
def test_reference_fixtures(ctx, name, force_exact):
    check_image(ctx, read_image(os.path.join(GOLD, "data", name)), force_exact)


@pytest.mark.parametrize("style", ["casava", "slash", "int", "nosuffix"])
@pytest.mark.parametrize("hdr2", [False, True])
@pytest.mark.parametrize("crlf", [False, True])
def test_clean_random_files(ctx, style, hdr2, crlf):
    rng = np.random.default_rng(hash((style, hdr2, crlf)) & 0xFFFF)
    img = fuzz.make_fastq(rng, 3000, 1, 300, style, hdr2, crlf)
    got = check_image(ctx, img)
    assert got["code"] == 0 and got["n_records"] == 3000
    assert got["consumed"] == len(img)


@pytest.mark.parametrize("kind", fuzz.MUTATIONS)
def test_mutated_files(ctx, kind):
    rng = np.random.default_rng(abs(hash(kind)) % 100000)
    for trial in range(12):
        style = ["casava", "slash", "int", "nosuffix"][trial % 4]
        img = fuzz.make_fastq(rng, int(rng.integers(1, 400)), 1, 120, style,
                              hdr2_names=bool(trial & 1), crlf=(trial % 5 == 4), rna=(trial % 6 == 5))
        for _ in range(int(rng.integers(1, 3))):
            img = fuzz.mutate(rng, img, kind)
        check_image(ctx, img)
        check_image(ctx, img, force_exact=True)


def test_edge_images(ctx):
    for img in [b"", b"\n", b"@", b"@a", b"@a\n", b"@a\nA", b"@a\nA\n+", b"@a\nA\n+\nI", b"@a\nA\n+\nI\n",
                b"\n\n\n\n", b"@a\nA\n+\nI\n\n", b"@a\nA\n+\nI\n@", b"\x00", b"@a\nA\n+\nI\n\x00@b\nA\n+\nI\n",
                b"@a\nA\n+\n\x00I\n", b"@a\n\x00A\n+\nI\n", b"@a\nA\n\x00+\nI\n", b"@a\x00b\nA\n+a\nI\n",
                b"@a b\nAC\x00GT\n+\nII\n", b"@a\nAC\rGT\n+\nII\n", b"@a\nACGT\n+\nII\rII\n",
                b"@\r\nA\n+\nI\n", b"@a/1\nA\n+a/1\nI\n", b"@a/1\nA\n+a/2\nI\n", b"@a/1\nA\n+a\nI\n"]:
        check_image(ctx, img)


def test_long_reads(ctx):
    rng = np.random.default_rng(7)
    img = fuzz.make_fastq(rng, 40, 5000, 60000, "nosuffix")
    got = check_image(ctx, img)
    assert got["n_records"] == 40


def test_streaming_chunks_carry(ctx):
    """final=0: an incomplete tail is carried, not an error; stats accumulate across images."""
    rng = np.random.default_rng(11)
    img = fuzz.make_fastq(rng, 2000, 20, 150, "casava")
    want = expected_from_oracle(img)
    st = fq.abi.probe_first_record(img, True)
    acc = ctx.accumulator()
    pos, total = 0, 0
    step = 37000
    carry = b""
    while pos < len(img):
        piece = carry + img[pos:pos + step]
        pos += step
        final = pos >= len(img)
        got = ctx.validate(piece, acc, st, final=final)
        assert got["code"] == 0
        total += got["n_records"]
        carry = piece[got["consumed"]:]
        if final:
            assert carry == b""
    s = want["summary"]
    stats = acc.read()
    assert total == s["num_reads"] == stats["num_rds"]
    assert (stats["min_rl"], stats["max_rl"]) == (s["min_rl"], s["max_rl"])
    assert (stats["min_qual"], stats["max_qual"]) == (s["min_qual"], s["max_qual"])
    assert acc.median() == s["median_rl"]
    acc.close()


def test_frame_records_descriptors(ctx):
    rng = np.random.default_rng(5)
    img = fuzz.make_fastq(rng, 500, 1, 90, "slash", hdr2_names=True)
    st = fq.abi.probe_first_record(img, True)
    acc = ctx.accumulator()
    got = ctx.validate(img, acc, st)
    recs = ctx.frame_records(0, got["n_records"])
    lines = img.split(b"\n")
    off = 0
    for r, d in enumerate(recs):
        l = [len(x) + 1 for x in lines[4 * r:4 * r + 4]]
        assert d["offset"] == off
        assert [d["hdr1_len"], d["seq_len"], d["hdr2_len"], d["qual_len"]] == l
        assert d["read_len"] == l[1]
        off += sum(l)
    acc.close()


def test_line_too_long_is_refused(ctx):
    img = b"@" + b"a" * 1200 + b"\nACGT\n+\nIIII\n"
    st = fq.abi.probe_first_record(img, True)
    got = ctx.validate(img, None, st, flags=fq.abi.VALIDATE_NO_STATS)
    assert got["code"] == 15 and got["record"] == 0


def test_synthetic_generator_is_valid_and_deterministic(ctx):
    torch = pytest.importorskip("torch")
    n, L = 20000, 150
    R = fq.abi.synth_record_bytes(L)
    assert R == 349
    a = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    b = torch.empty(n * R, dtype=torch.uint8, device="cuda")
    ctx.synth_fastq(a.data_ptr(), n, L, first_index=0, seed=12345)
    ctx.synth_fastq(b.data_ptr(), n // 2, L, first_index=n // 2, seed=12345)
    ctx.synchronize()
    assert torch.equal(a[(n // 2) * R:], b[: (n // 2) * R])
    img = bytes(a.cpu().numpy())
    got = check_image(ctx, img)
    assert got["n_records"] == n
    st = fq.abi.probe_first_record(img[:1000], True)
    assert st.readname_format == fq.abi.NAME_CASAVA18
    # device-resident input gives the same answer as the host copy
    acc = ctx.accumulator()
    got_d = ctx.validate(a.data_ptr(), acc, st, nbytes=n * R)
    assert got_d["code"] == 0 and got_d["n_records"] == n
    acc.close()


FAST_KINDS = ["flip_seq", "del_byte", "drop_line", "dup_line", "bad_plus", "bad_at", "short_qual", "mix_ut",
              "empty_seq", "high_qual", "hdr2_name", "empty_hdr", "strip_last_nl", "truncate"]


@pytest.mark.parametrize("kind", FAST_KINDS)
def test_tiled_path_on_multi_tile_images(ctx, kind):
    """Images of many 16 KiB tiles with ONE defect somewhere: the tiled path (path == 2) must
    find exactly what the serial oracle finds, including defects on tile seams."""
    rng = np.random.default_rng(abs(hash("big" + kind)) % 100000)
    for trial in range(6):
        style = ["casava", "slash", "int", "nosuffix"][trial % 4]
        img = fuzz.make_fastq(rng, 6000, 20, 200, style)
        img = fuzz.mutate(rng, img, kind)
        got = check_image(ctx, img)
        assert got["path"] in (2, 3)
        check_image(ctx, img, force_exact=True)


def test_tiled_path_defects_on_every_seam_offset(ctx):
    """Slide one bad base / one bad '+' / one bad '@' across a tile boundary byte by byte."""
    rng = np.random.default_rng(3)
    base = bytearray(fuzz.make_fastq(rng, 400, 100, 100, "casava"))
    lines = bytes(base).split(b"\n")
    # record geometry is fixed (names differ in length a little): find records around 16 KiB
    offs, p = [], 0
    for ln in lines[:-1]:
        offs.append(p)
        p += len(ln) + 1
    seam = 16384
    near = [i for i, o in enumerate(offs) if seam - 400 < o < seam + 400]
    for li in near:
        for delta, repl in ((0, b"X"), (1, b"\n")):
            img = bytearray(base)
            pos = offs[li] + delta
            if pos < len(img) and img[pos:pos + 1] != b"\n":
                img[pos:pos + 1] = repl
                check_image(ctx, bytes(img))


def test_tiled_path_many_suspects_overflow_queue(ctx):
    """Every record repeats its name on line 3: each one is queued for the exact validator."""
    rng = np.random.default_rng(21)
    img = fuzz.make_fastq(rng, 30000, 30, 60, "slash", hdr2_names=True)
    got = check_image(ctx, img)
    assert got["path"] in (2, 3) and got["code"] == 0
    bad = img.replace(b"\n+read.29000/1\n", b"\n+read.29000/2x\n")
    got = check_image(ctx, bad)
    assert got["code"] == 10 and got["record"] == 29000


def test_lowercase_and_rna_take_the_exact_checks(ctx):
    rng = np.random.default_rng(8)
    img = fuzz.make_fastq(rng, 5000, 50, 120, "casava", rna=True)
    got = check_image(ctx, img)
    assert got["code"] == 0
    img2 = fuzz.make_fastq(rng, 5000, 50, 120, "casava").lower().replace(b"syn:", b"SYN:")
    check_image(ctx, img2)
