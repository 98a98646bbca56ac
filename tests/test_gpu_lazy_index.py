"""The line index on demand (include/fqg.h: FQG_VALIDATE_INDEX): a call of fqg_validate that takes the single-pass
framing path and is not told that its frame will be used stores only the end of the index, and whatever needs the
frame afterwards - fqg_frame_records, fqg_frame_retain, the name calls, the call's own exact validator for records the
line kernels could not vouch for - has the rest written first.  Same results either way: every case here runs both ways
(and through the two-pass path) and compares everything a caller can see, the record offsets included."""
import zlib

import numpy as np
import pytest

from tests import fuzz
from tests.test_gpu_validate import check_image

pytestmark = pytest.mark.gpu
fq = pytest.importorskip("fastq_utils_amd")
A = fq.abi


@pytest.fixture(scope="module")
def ctx():
    c = fq.Context(0)
    yield c
    c.close()


def both_ways(ctx, img, final=True, records_at=(0, 1, 7), want_stream=True):
    """-> the on-demand result; asserts that the eager and the two-pass runs say the same, and that the records read
    back from the frame are the same in all three"""
    st = A.probe_first_record(img, True)
    seen = []
    for flags in (0, A.VALIDATE_INDEX, A.VALIDATE_TWO_PASS):
        acc = ctx.accumulator()
        r = ctx.validate(img, acc, st, final=final, flags=flags)
        n = r["n_records"]
        recs = []
        if n:
            firsts = sorted({min(x, n - 1) for x in records_at} | {n - 1, n // 2})
            recs = [ctx.frame_records(f, min(3, n - f)) for f in firsts]
        summary = (acc.read(), acc.hist())
        acc.close()
        seen.append(({k: v for k, v in r.items() if k != "path"}, recs, summary, r["path"]))
    assert seen[0][3] == seen[1][3] and seen[2][3] in (1, 2), [s[3] for s in seen]
    if len(img) >= (1 << 20) and want_stream:
        assert seen[0][3] == 3
    for other in seen[1:]:
        assert seen[0][0] == other[0]
        assert seen[0][1] == other[1]
        assert seen[0][2] == other[2]
    return seen[0][0]


def clean(rng, n=9000, lo=80, hi=160, style="casava"):
    img = fuzz.make_fastq(rng, n, lo, hi, style)
    assert len(img) > (1 << 20)
    return img


def test_clean_image_records_read_back(ctx):
    r = both_ways(ctx, clean(np.random.default_rng(11)))
    assert r["code"] == 0 and r["n_records"] == 9000


def test_fixed_length_reads_take_the_kernel_without_a_search(ctx):
    # (32 - 60 newlines per 4 KiB chunk: k_stream_lines_fast and the steps it marks)
    r = both_ways(ctx, clean(np.random.default_rng(12), n=12000, lo=150, hi=150), records_at=(0, 63, 64, 127, 128, 5000))
    assert r["code"] == 0 and r["n_records"] == 12000


@pytest.mark.parametrize("kind", ["short_qual", "bad_plus", "bad_at", "empty_hdr", "flip_seq", "hdr2_name", "empty_seq",
                                  "drop_line", "dup_line", "truncate", "strip_last_nl", "del_byte"])
def test_one_defect(ctx, kind):
    """a record the line kernels cannot vouch for (lengths, '+' line, '@') goes to the exact validator, which reads it
    through the index: written on demand inside the call; defects pass 1 queues keep the index in the call"""
    rng = np.random.default_rng(zlib.crc32(("lazy" + kind).encode()))
    for trial in range(2):
        img = fuzz.mutate(rng, clean(rng, lo=[80, 150][trial], hi=[160, 150][trial]), kind)
        got = both_ways(ctx, img)
        want = check_image(ctx, img)   # (the oracle's verdict, through the default flags)
        assert got["code"] == want["code"] and got["record"] == want["record"]


@pytest.mark.parametrize("tail", [b"", b"@x", b"@x\nAC", b"@x\nAC\n+", b"@x\nAC\n+\nII", b"@x\nAC\n+\nI"])
def test_incomplete_last_record_and_a_piece_that_is_not_the_last(ctx, tail):
    img = clean(np.random.default_rng(13), n=8000, lo=150, hi=150) + tail
    for final in (True, False):
        r = both_ways(ctx, img, final=final)
        assert r["n_records"] >= 8000


def test_step_boundaries_of_the_line_kernels(ctx):
    """the records kept in the call are the last two; whole steps of 128 records end exactly there or just before"""
    rng = np.random.default_rng(14)
    for n in (128 * 40, 128 * 40 + 1, 128 * 40 + 2, 128 * 40 - 1, 128 * 40 + 127):
        img = fuzz.make_fastq(rng, n, 150, 150, "casava")
        if len(img) < (1 << 20):
            img = fuzz.make_fastq(rng, n + 128 * 10, 150, 150, "casava")
        both_ways(ctx, img, records_at=(0, 126, 127, 128, 129))


def test_retained_frame_and_names_after_an_on_demand_call(ctx):
    rng = np.random.default_rng(15)
    img = clean(rng, n=9000, lo=150, hi=150)
    st = A.probe_first_record(img, True)
    out = []
    for flags in (0, A.VALIDATE_INDEX):
        r = ctx.validate(img, None, st, flags=flags | A.VALIDATE_NO_STATS)
        fr = ctx.retain_frame()
        names = [ctx.frame_name(fr, st, k) for k in (0, 1, 63, 64, 127, 128, 4500, 8998, 8999)]
        cmp_ = ctx.names_compare(fr, st)
        out.append((r["n_records"], r["consumed"], names, cmp_))
        fr.release()
    assert out[0] == out[1] and out[0][0] == 9000


def test_the_name_index_after_an_on_demand_call(ctx):
    """fqg_index_insert_unique on the frame a call WITHOUT FQG_VALIDATE_NAMES left: the names come through the line
    index, which is written first"""
    rng = np.random.default_rng(17)
    img = clean(rng, n=9000, lo=150, hi=150)
    st = A.probe_first_record(img, True)
    got = []
    for flags in (0, A.VALIDATE_INDEX, A.VALIDATE_NAMES):
        ctx.validate(img, None, st, flags=flags | A.VALIDATE_NO_STATS)
        idx = ctx.name_index(9000)
        got.append(idx.insert_unique(st))
        idx.close()
    assert got[0] == got[1] == got[2]


def test_a_second_call_forgets_the_first_calls_index(ctx):
    rng = np.random.default_rng(16)
    a, b = clean(rng, n=9000), clean(rng, n=7000)
    st = A.probe_first_record(a, True)
    ctx.validate(a, None, st, flags=A.VALIDATE_NO_STATS)
    ctx.validate(b, None, st, flags=A.VALIDATE_NO_STATS)
    got = ctx.frame_records(6990, 10)
    ctx.validate(b, None, st, flags=A.VALIDATE_NO_STATS | A.VALIDATE_TWO_PASS)
    assert got == ctx.frame_records(6990, 10)


def test_fingerprints_of_the_current_frame_after_an_on_demand_call(ctx):
    """fqg_names_fingerprints on the CURRENT frame (no retained one): the export reads the names through the line index"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(18)
    img = clean(rng, n=9000, lo=150, hi=150)
    st = A.probe_first_record(img, False)
    got = []
    for flags in (0, A.VALIDATE_INDEX):
        r = ctx.validate(img, None, st, flags=flags | A.VALIDATE_NO_STATS)
        buf = torch.zeros(r["n_records"] * 16, dtype=torch.uint8, device="cuda")
        counts = ctx.names_fingerprints(None, st, 0, 3, buf.data_ptr())
        ctx.synchronize()
        pairs = buf.cpu().numpy().view(np.uint64).reshape(-1, 2)
        start, buckets = 0, []
        for c in counts:  # (the order inside a bucket is not fixed: sorted)
            b = pairs[start:start + c]
            buckets.append(b[np.lexsort((b[:, 1], b[:, 0]))].tobytes())
            start += c
        got.append((counts, buckets))
    assert got[0] == got[1] and sum(got[0][0]) == 9000
