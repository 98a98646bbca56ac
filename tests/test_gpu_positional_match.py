"""Paired mode, file 2: an asker first looks at the record of file 1 at ITS OWN place (match_name,
fqg_index_kernels.hip: IndexView::n_positional) - mate files hold their reads in one order, and the name record next to
the neighbours' is a sequential read where the table is a random one.  The answer must be the table's: every arrangement
here is run with the shortcut and without it (FQGPU_NO_POSITIONAL_MATCH=1), in pieces of 1 MiB (askers of a later piece,
claims that meet across the two ways), against the oracle's serial loop (src/fastq_info.c:197-260: look the name up,
delete the entry, stop at the first name that is not there; entries left at the end are unpaired reads)."""
import os
import tempfile

import numpy as np
import pytest

from tests.test_gpu_cli import compare_with_oracle
from tests.test_gpu_name_paths import casava, rec, write

pytestmark = pytest.mark.gpu
N = 9000


def mates(rng, n=N):
    return [rec(*casava(i, 1), rng) for i in range(n)], [rec(*casava(i, 2), rng) for i in range(n)]


def arrangements():
    rng = np.random.default_rng(5)
    a, b = mates(rng)
    out = {"ordered": b}
    out["shifted_by_one"] = b[1:] + b[:1]                       # every asker goes through the table
    half = N // 2
    out["second_half_shuffled"] = b[:half] + [b[i] for i in half + rng.permutation(N - half)]
    out["two_swapped"] = b[:100] + [b[4000]] + b[101:4000] + [b[100]] + b[4001:]
    # a name asked for twice: at its own place and, later / earlier, at another one (the claims of the two ways meet)
    out["twice_later"] = b[:7000] + [b[5]] + b[7001:]           # serial loop: record 7000 finds nothing
    out["twice_earlier"] = b[:5] + [b[7000]] + b[6:]            # ... record 7000 finds its entry taken (by record 5)
    out["one_missing"] = b[:3000] + b[3001:]                    # file 1 keeps an entry
    out["one_unknown"] = b[:3000] + [rec(*casava(N + 77, 2), rng)] + b[3001:]
    out["short"] = b[:6000]
    return [x for x in a], out


@pytest.mark.parametrize("contexts", ["one", "0,0", "0,0,0"])
@pytest.mark.parametrize("shortcut", ["on", "off"])
def test_the_shortcut_gives_the_tables_answer(shortcut, contexts):
    """contexts "0,0" / "0,0,0": the files spread over several contexts (FQGPU_DEVICES, host/fq_names_multi.h) - there
    the shortcut compares the name records of the two files range by range (fqg_frame_name_records /
    fqg_frame_names_equal) and the exchange by hash runs only when a name is not at its place"""
    a, files2 = arrangements()
    files = {"a_1.fastq": b"".join(a)}
    for k, v in files2.items():
        files[k + "_2.fastq"] = b"".join(v)
    env = {"FQGPU_CHUNK_MB": "1"}
    if contexts != "one":
        env["FQGPU_DEVICES"] = contexts
    if shortcut == "off":
        env["FQGPU_NO_POSITIONAL_MATCH"] = "1"
    with tempfile.TemporaryDirectory() as tmp:
        write(tmp, files)
        for k in files2:
            compare_with_oracle(tmp, ["a_1.fastq", k + "_2.fastq"], files, env)


def test_a_file_1_with_a_repeated_name_turns_the_shortcut_off():
    """the ABI lets a caller go on after fqg_index_insert_unique has reported a repeat (the program stops there): the
    table then holds that name once, at its FIRST record, and the second record's own place must not stand in for it"""
    import fastq_utils_amd as fq
    A = fq.abi
    rng = np.random.default_rng(8)
    a, b = mates(rng, 3000)
    a[2000] = rec(*casava(10, 1), rng)           # record 2000 repeats the name of record 10
    b[2000] = rec(*casava(10, 2), rng)           # ... and so does file 2: askers 10 and 2000 want the same entry
    img1, img2 = b"".join(a), b"".join(b)
    with fq.Context(0) as ctx:
        st1 = A.probe_first_record(img1, True)
        r = ctx.validate(img1, None, st1, flags=A.VALIDATE_NO_STATS | A.VALIDATE_NAMES)
        assert r["code"] == 0
        idx = ctx.name_index(4000)
        ir = idx.insert_unique(st1)
        assert ir["code"] != 0 and ir["record"] == 2000
        st2 = A.probe_first_record(img2, True)
        ctx.validate(img2, None, st2, flags=A.VALIDATE_NO_STATS | A.VALIDATE_NAMES)
        m = idx.match_delete(st2)
        # asker 10 takes the entry, asker 2000 finds it gone (with the shortcut on it would have found "its" record 2000)
        assert m["code"] != 0 and m["record"] == 2000, m
        idx.close()


def test_name_records_and_their_comparison():
    """the two calls of the comparison by position on ranges of two frames: counts against a recount on the host"""
    import ctypes as C

    import fastq_utils_amd as fq
    A = fq.abi
    L = A.load()
    rng = np.random.default_rng(3)
    a, b = mates(rng, 5000)
    b[1234] = rec(*casava(4321, 2), rng)                                  # another read's name
    b[2000] = rec(b"SRX:7:FC9:1:1:1:" + b"9" * 70, b" 2:N:0:ACGTAC", rng)   # 86 bytes: beyond what a record holds ...
    a[2000] = rec(b"SRX:7:FC9:1:1:1:" + b"9" * 70, b" 1:N:0:ACGTAC", rng)   # ... and equal: undecided
    b[3000] = b"X" + b[3000][1:]                                          # no '@'
    img1, img2 = b"".join(a), b"".join(b)
    with fq.Context(0) as ctx:
        frames, states = [], []
        for img in (img1, img2):
            st = A.probe_first_record(img, True)
            ctx.validate(img, None, st, flags=A.VALIDATE_FRAME_ONLY)
            frames.append(ctx.retain_frame())
            states.append(st)
        for first, n in ((0, 5000), (1000, 500), (1234, 1), (1990, 20), (2999, 3), (4999, 1), (17, 0)):
            buf = L.fqg_device_alloc(ctx.h, max(n, 1) * 64)
            assert buf
            ctx._check(L.fqg_frame_name_records(ctx.h, frames[0].h, C.byref(states[0]), first, n, buf))
            eq, und = C.c_uint64(0), C.c_uint64(0)
            ctx._check(L.fqg_frame_names_equal(ctx.h, frames[1].h, C.byref(states[1]), first, n, buf, C.byref(eq), C.byref(und)))
            L.fqg_device_free(ctx.h, buf)
            bad = sum(1 for x in (1234, 2000, 3000) if first <= x < first + n)
            assert (eq.value, und.value) == (n - bad, 1 if first <= 2000 < first + n else 0), (first, n, eq.value, und.value)
        # beyond the frame: refused
        buf = L.fqg_device_alloc(ctx.h, 64)
        assert L.fqg_frame_name_records(ctx.h, frames[0].h, C.byref(states[0]), 4999, 2, buf) != 0
        L.fqg_device_free(ctx.h, buf)
        for f in frames:
            f.release()
