"""GPU parity tests aimed at the single-pass (streaming) framing path of fqg_validate: images
large enough to take it by default (>= 1 MiB), and inputs built to defeat its speculation - no
"+" line inside a 4 KiB chunk, one-character lines that are not "+" lines, quality bytes outside
the range seen at the start of the file, control / high bytes far into the image, more newlines
per chunk than the staging area holds, more suspects than the queue holds.  The oracle
(oracle/fq_oracle.c, `fastq_info -r` semantics) is the checker; results must be identical."""
import numpy as np
import pytest

from tests import fuzz
from tests.test_gpu_validate import check_image

pytestmark = pytest.mark.gpu
fq = pytest.importorskip("fastq_utils_amd")


@pytest.fixture(scope="module")
def ctx():
    c = fq.Context(0)
    yield c
    c.close()


def big_clean(rng, n=9000, lo=80, hi=160, style="casava"):
    img = fuzz.make_fastq(rng, n, lo, hi, style)
    assert len(img) > (1 << 20)
    return img


def test_default_threshold_takes_the_streaming_path(ctx):
    rng = np.random.default_rng(1)
    img = big_clean(rng)
    got = check_image(ctx, img)
    assert got["path"] == 3 and got["code"] == 0
    st = fq.abi.probe_first_record(img, True)
    acc = ctx.accumulator()
    two = ctx.validate(img, acc, st, flags=fq.abi.VALIDATE_TWO_PASS)
    assert two["path"] == 2 and two["n_records"] == got["n_records"] and two["consumed"] == got["consumed"]
    acc.close()


@pytest.mark.parametrize("kind", ["flip_seq", "bad_plus", "bad_at", "short_qual", "empty_hdr", "hdr2_name",
                                  "drop_line", "dup_line", "del_byte", "truncate", "strip_last_nl", "empty_seq",
                                  "mix_ut", "high_qual"])
def test_one_defect_in_a_large_image(ctx, kind):
    rng = np.random.default_rng(abs(hash("stream" + kind)) % 100000)
    for trial in range(3):
        img = fuzz.mutate(rng, big_clean(rng, style=["casava", "slash", "int"][trial]), kind)
        got = check_image(ctx, img)
        if kind != "high_qual":  # (that one may plant a byte >= 0x80, which the streaming path hands back)
            assert got["path"] == (3 if len(img) >= (1 << 20) else 2)


@pytest.mark.parametrize("byte", [0, 13, 9, 27, 0x80, 0xFF])
def test_control_or_high_byte_far_into_the_image(ctx, byte):
    """A NUL / CR / byte >= 0x80 anywhere makes the image ineligible for the streaming path; any
    other control byte (a tab in a header ...) is just a byte."""
    rng = np.random.default_rng(byte + 5)
    img = bytearray(big_clean(rng))
    for where in ("hdr", "seq", "qual"):
        b = bytearray(img)
        lines_at = 700000
        p = b.index(b"\n@", lines_at) + 1          # start of a record
        if where == "hdr":
            p += 3
        elif where == "seq":
            p = b.index(b"\n", p) + 5
        else:
            p = b.index(b"\n+\n", p) + 8
        b[p] = byte
        got = check_image(ctx, bytes(b))
        assert got["path"] == (3 if byte in (9, 27) else 2 if byte >= 0x80 else 1)


def test_long_reads_have_no_plus_line_in_most_chunks(ctx):
    rng = np.random.default_rng(2)
    img = fuzz.make_fastq(rng, 60, 9000, 40000, "nosuffix")
    assert len(img) > (1 << 20)
    got = check_image(ctx, img)
    assert got["path"] == 3 and got["n_records"] == 60
    bad = bytearray(img)
    p = bad.index(b"\n", bad.index(b"\n@", 900000) + 1) + 12345 % 9000
    bad[p] = ord("x")
    check_image(ctx, bytes(bad))


def test_one_base_reads_mislead_the_speculation(ctx):
    """Reads of length 1: sequence, "+" and quality lines are all one byte long, so the first
    one-byte line of a chunk is usually not the "+" line."""
    rng = np.random.default_rng(3)
    parts = []
    for i in range(40000):
        L = 1 if i % 3 else int(rng.integers(2, 40))
        seq = fuzz.BASES[rng.integers(0, 4, L)].tobytes()
        qual = bytes((rng.integers(2, 41, L) + 33).astype(np.uint8))
        parts.append(b"@r%d%s\n%s\n+\n%s\n" % (i, b"_" * 100 if i != 33334 else b"", seq, qual))
    img = b"".join(parts)
    assert len(img) > (1 << 20)
    got = check_image(ctx, img)
    assert got["path"] == 3 and got["n_records"] == 40000
    k = img.index(b"@r33334\n") + len(b"@r33334\n")
    bad = img[:k] + b"#" + img[k + 1:]
    got = check_image(ctx, bad)
    assert got["code"] == 6 and got["record"] == 33334


def test_quality_bytes_outside_the_range_seen_at_the_start(ctx):
    rng = np.random.default_rng(4)
    img = bytearray(big_clean(rng))
    # one very low and one very high quality byte late in the file
    q1 = img.index(b"\n+\n", 500000) + 3 + 7
    q2 = img.index(b"\n+\n", 900000) + 3 + 2
    img[q1] = ord("!")
    img[q2] = ord("~")
    got = check_image(ctx, bytes(img))
    assert got["path"] == 3 and got["code"] == 0
    # the same inside a chunk whose speculation is wrong (one-base read right at its start)
    rec = b"@one\nA\n+\n!\n"
    k = (img.index(b"\n@", 655360) + 1)
    img2 = bytes(img[:k]) + rec + bytes(img[k:])
    check_image(ctx, img2)


def test_short_lines_overflow_the_staging_area(ctx):
    """More than 256 newlines in a 4 KiB chunk: the image goes back to the two-pass path."""
    rng = np.random.default_rng(5)
    img = fuzz.make_fastq(rng, 120000, 1, 3, "int")
    assert len(img) > (1 << 20)
    got = check_image(ctx, img)
    assert got["path"] == 2 and got["n_records"] == 120000


def test_more_suspects_than_the_queue_holds(ctx):
    """Every sequence line carries a lower-case base: > 2^20 suspect positions."""
    rng = np.random.default_rng(6)
    parts = []
    for i in range(1100000):
        parts.append(b"@%d\nAcGT\n+\nIIII\n" % i)
    img = b"".join(parts)
    got = check_image(ctx, img)
    assert got["code"] == 0 and got["n_records"] == 1100000
    bad = img.replace(b"@1000000\nAcGT", b"@1000000\nAcGX", 1)
    got = check_image(ctx, bad)
    assert got["code"] == 6 and got["record"] == 1000000


def test_streamed_pieces_with_carry(ctx):
    rng = np.random.default_rng(8)
    img = big_clean(rng, n=30000)
    st = fq.abi.probe_first_record(img, True)
    from tests.test_gpu_validate import expected_from_oracle
    want = expected_from_oracle(img)["summary"]
    acc = ctx.accumulator()
    pos, total, carry, step = 0, 0, b"", (1 << 20) + 12345
    while pos < len(img):
        piece = carry + img[pos:pos + step]
        pos += step
        final = pos >= len(img)
        got = ctx.validate(piece, acc, st, final=final)
        assert got["code"] == 0 and got["path"] == (3 if len(piece) >= (1 << 20) else 2)
        total += got["n_records"]
        carry = piece[got["consumed"]:]
    s = acc.read()
    assert total == want["num_reads"] == s["num_rds"]
    assert (s["min_rl"], s["max_rl"], s["min_qual"], s["max_qual"]) == (
        want["min_rl"], want["max_rl"], want["min_qual"], want["max_qual"])
    assert acc.median() == want["median_rl"]
    acc.close()
