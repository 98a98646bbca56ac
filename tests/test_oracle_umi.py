"""Pin the bam_umi_count oracle (oracle/umi_oracle.py) on the golden vectors captured from the
reference binary (tests/golden/umi_count.json, tools/gen_golden.py umi): the reference suite's own
invocations on its BAM fixtures (run_tests.sh:96-177) plus seeded synthetic BAMs (tests/bamgen.py)
that exercise NH weights, multi-gene GX tags, missing tags, whitelists and both output modes."""
import json
import os

import pytest

from oracle import umi_oracle as uo
from tests.util import GOLD

GOLDEN = json.load(open(os.path.join(GOLD, "umi_count.json")))


def reader(path):
    full = os.path.join(GOLD, path)
    if not os.path.exists(full):
        return None
    with open(full, "rb") as f:
        return f.read()


def real_args(args):
    return [a.replace("OUTU", "SCRATCH/u.mtx").replace("OUTR", "SCRATCH/r.mtx") for a in args]


def want_exit(case):
    return {-6: 134}.get(case["exit"], case["exit"])  # SIGABRT from assert() as a shell reports it


def comparable(case):
    """Left out: invocations on which the reference crashes with SIGSEGV (array overruns when
    --max_feat is smaller than the number of features: no defined output)."""
    return case["exit"] != -11


def golden_files(case):
    out = {}
    for k, v in case["files"].items():
        tag, _, ext = k.partition("_")
        base = "SCRATCH/u.mtx" if tag == "OUTU" else "SCRATCH/r.mtx"
        out[base + ("_" + ext if ext else "")] = v
    return out


@pytest.mark.parametrize("case", [c for c in GOLDEN if comparable(c)],
                         ids=lambda c: " ".join(c["args"])[:80])
def test_oracle_matches_reference_binary(case):
    got = uo.run_bam_umi_count(real_args(case["args"]), reader)
    assert got["exit"] == want_exit(case)
    if case["exit"] == -6:
        assert "Assertion `len1+1 < FEAT_ID_MAX_LEN' failed" in got["stderr"]
        return
    assert got["stderr"] == case["stderr"]
    if case["exit"] == 0:
        assert got["files"] == golden_files(case)


def test_reference_suite_known_answers():
    """run_tests.sh:117 (89 lines without the % line), :153 (365 columns), :158 (4 lines)"""
    by = {" ".join(c["args"]): c for c in GOLDEN}
    c = by["--min_reads 1 --bam data_umi/test_annot5.bam --multi_mapped --ucounts OUTU --not_sorted_by_cell"]
    assert len([ln for ln in c["files"]["OUTU"].splitlines() if "%" not in ln]) == 89
    c = by["--min_reads 1 --bam data_umi/test_annot5.bam --ucounts OUTU --ignore_sample --not_sorted_by_cell "
           "--cell_suffix -123456789"]
    assert c["files"]["OUTU_cols"].count("123456789") == 365
    c = by["--not_sorted_by_cell --min_reads 1 --bam data_umi/test_annot5.bam --known_cells data_umi/known_cells.txt "
           "--ucounts OUTU"]
    assert len(c["files"]["OUTU"].splitlines()) == 4
    c = by["--bam data_umi/test_one_cell.bam --ucounts OUTU"]
    assert c["files"]["OUTU"].splitlines()[1].split() == ["40", "1", "3257"]  # SURVEY.md 8c [probe]


def test_reference_rl_tree_defect():
    """The reference keeps the UMIs of a (cell, gene) in an RL_Tree (src/range_list.c).  Inserting a
    number whose new node lands directly in front of the last node of the array drops that last node
    (shift_right() moves nothing for one trailing node, :287-301, called from new_node :338-339), and
    rl_all(OUT) between cells leaves stale nodes behind (:187-198): members are lost or invented when
    UMI ids do not arrive in increasing order.  The oracle restates the tree as it behaves
    (oracle/rl_oracle.c); this pins it on the smallest input that shows the defect and on the larger
    seeded inputs with re-used UMIs - all byte-identical to the reference binary."""
    by = {" ".join(c["args"]): c for c in GOLDEN}
    c = by["--bam data_umi/rl_defect.bam --ucounts OUTU"]
    ref = [ln.split() for ln in c["files"]["OUTU"].splitlines()[2:]]
    assert ref == [["1", "1", "40"], ["2", "1", "3"]]          # the reference: 3 "distinct" UMIs on gene B
    got = uo.run_bam_umi_count(real_args(c["args"]), reader)   # (a set would hold 2: UMI 20, UMI 40)
    assert got["files"] == golden_files(c)
    assert got["overwrites"] >= 1
    for key in ("--bam data_umi/syn_reuse.bam --ucounts OUTU --rcounts OUTR",
                "--bam data_umi/syn_reuse.bam --ucounts OUTU --rcounts OUTR --not_sorted_by_cell"):
        c = by[key]
        got = uo.run_bam_umi_count(real_args(c["args"]), reader)
        assert got["exit"] == c["exit"] == 0
        assert got["files"] == golden_files(c)


UNSORTED = [[1, 2, 3, 1], [1, 2, 1, 3], [1, 1, 2, 3, 4, 2, 5], [1, 2, 3, 4, 5, 6, 7, 8, 3], [1, 2, 2, 1]]


@pytest.mark.parametrize("order", UNSORTED, ids=lambda o: "".join(map(str, o)))
@pytest.mark.parametrize("extra", [[], ["--min_reads", "2"], ["--min_umis", "2"]], ids=["plain", "min_reads", "min_umis"])
def test_a_bam_that_is_not_grouped_by_cell_against_the_reference_binary(order, extra, tmp_path):
    """src/bam_umi_count.c:1000-1015: the reference stops at the alignment that is out of order with the complete cells
    in its files (all but the one in front of that alignment), behind the header it never comes back to"""
    import subprocess

    import numpy as np

    from tests import bamgen
    from tests.util import REPO
    ref = os.path.join(REPO, "oracle", "_ref", "bam_umi_count")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/bam_umi_count not built")
    bam = bamgen.bgzf(bamgen.cells_in_runs(np.random.default_rng(sum(order) * 7 + len(order)), order), level=1)
    (tmp_path / "in.bam").write_bytes(bam)
    args = ["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"] + extra
    p = subprocess.run(["bam_umi_count"] + args, executable=ref, cwd=tmp_path, capture_output=True, timeout=300)
    got = uo.run_bam_umi_count(args, lambda path: bam if path == "in.bam" else None)
    assert p.returncode == 1 and got["exit"] == 1
    assert got["stderr"] == p.stderr.decode("latin-1")
    for name in ("u", "r"):
        assert got["files"].get(name) == (tmp_path / name).read_text(), name
    assert sorted(got["files"]) == ["r", "u"] and sorted(x.name for x in tmp_path.iterdir()) == ["in.bam", "r", "u"]
