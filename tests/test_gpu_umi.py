"""GPU parity tests for the bam_umi_count path: bin/bam_umi_count (C++ host + libfqgpu.so) against
every comparable golden invocation of the reference binary (tests/golden/umi_count.json), and
fqg_umi_count through the C-ABI against the oracle (oracle/umi_oracle.py) on seeded BAMs with
re-used UMIs, NH weights, multi-gene tags, whitelists, in both output modes.  Integer / byte work
and float32 counters added in record order: the bar is bit-exact."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import umi_oracle as uo
from tests import bamgen
from tests.test_oracle_umi import GOLDEN, comparable, golden_files, want_exit
from tests.util import GOLD, REPO, SideBySide, free_port

pytestmark = pytest.mark.gpu
fq = pytest.importorskip("fastq_utils_amd")
BIN = os.path.join(REPO, "bin", "bam_umi_count")


def run_cli(args, cwd):
    # argv[0] as in the golden runs: getopt's own messages start with it
    p = subprocess.run(["bam_umi_count"] + args, executable=BIN, cwd=cwd, capture_output=True, timeout=300)
    return p.returncode, p.stderr.decode("latin-1")


CLI_CASES = [c for c in GOLDEN if comparable(c)]


def cli_case_run(i):
    case = CLI_CASES[i]
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        real = [a.replace("OUTU", rel + "/u.mtx").replace("OUTR", rel + "/r.mtx") for a in case["args"]]
        rc, err = run_cli(real, GOLD)
        got = {}
        for base in ("u.mtx", "r.mtx"):
            for ext in ("", "_rows", "_cols"):
                path = os.path.join(tmp, base + ext)
                if os.path.exists(path):
                    got["SCRATCH/" + base + ext] = open(path, "rb").read().decode("latin-1")
    return rc, err.replace(rel + "/", "SCRATCH/"), got


# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
CLI_RUNS = SideBySide(cli_case_run, range(len(CLI_CASES)))


@pytest.mark.parametrize("i", range(len(CLI_CASES)), ids=[" ".join(c["args"])[:80] for c in CLI_CASES])
def test_cli_matches_reference_binary(i):
    case = CLI_CASES[i]
    rc, err, got = CLI_RUNS.get(i)
    assert (rc if rc >= 0 else 128 - rc) == want_exit(case), err[-400:]
    if case["exit"] == -6:
        assert "Assertion `len1+1 < FEAT_ID_MAX_LEN' failed" in err
        return
    assert err == case["stderr"]
    if case["exit"] == 0:
        assert got == golden_files(case)


@pytest.fixture(scope="module")
def ctx():
    c = fq.Context(0)
    yield c
    c.close()


def oracle_files(bam, extra):
    files = {"in.bam": bam}
    got = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"] + extra, lambda p: files.get(p))
    return got


def lines_of(text):
    return [tuple(int(x) for x in ln.split()) for ln in text.splitlines()[2:]]


@pytest.mark.parametrize("variant", ["plain", "nh", "multi", "noise", "unsorted", "whitelist", "uniq", "thresholds"])
def test_abi_matches_oracle_on_reused_umis(ctx, variant):
    """UMIs re-used across genes and cells: the regime in which the reference's RL_Tree loses and invents
    members (tests/test_oracle_umi.py::test_reference_rl_tree_defect).  The oracle restates the tree as it
    behaves; the HIP path detects the affected (cell, gene) sets and replays them (fqg_rl_sim.h)."""
    rng = np.random.default_rng(abs(hash(variant)) % 9999)
    for trial in range(3):
        kw = dict(n_cells=int(rng.integers(3, 60)), genes=int(rng.integers(5, 400)), umi_len=int(rng.integers(2, 9)),
                  reads_per_cell=(1, int(rng.integers(2, 300))))
        if variant in ("nh", "multi", "noise", "uniq"):
            kw["nh"] = True
        if variant in ("multi", "noise"):
            kw["multi_gx"] = True
        if variant == "noise":
            kw["noise"] = True
        sorted_mode = variant != "unsorted"
        if not sorted_mode:
            kw["sort_cells"] = False
        bam, stream = bamgen.tagged_bam(rng, **kw)
        extra, args = [], {}
        if not sorted_mode:
            extra.append("--not_sorted_by_cell")
        if variant == "uniq":
            extra.append("--uniq_mapped")
            args["uniq_mapped_only"] = True
        if variant == "thresholds":
            extra += ["--min_reads", "2", "--min_umis", "2"]
            args.update(min_reads=2, min_umis=2)
        files = {"in.bam": bam}
        if variant == "whitelist":
            cells = []
            for tid, flag, aux in uo.bam_records(stream):
                c = uo.get_tag(aux, b"CR")
                if c and c not in cells:
                    cells.append(c)
            keep = cells[::2]
            files["wl.txt"] = b"".join(c + b"\n" for c in keep)
            extra += ["--known_cells", "wl.txt"]
            args["known_cells"] = [uo.char2uint_64(c) for c in keep]
        want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"] + extra, files.get)
        got = ctx.umi_count(stream, sorted_by_cell=sorted_mode, **args)
        assert want["exit"] == 0 and got["code"] == 0
        assert got["entries"][0] == lines_of(want["files"]["u"])
        assert got["entries"][1] == lines_of(want["files"]["r"])
        rows = [ln.split("\t")[1] for ln in want["files"]["u_rows"].splitlines()]
        assert [f.decode() for f in got["features"]] == rows
        cols = [ln.split("\t")[1] for ln in want["files"]["u_cols"].splitlines()]
        assert [uo.uint_642char(c).decode() for c in got["cells"]] == cols
        tail = want["stderr"].splitlines()
        assert "%f total reads" % got["tot_reads"] in tail and "%f total UMI" % got["tot_umi"] in tail
        if sorted_mode:
            hdr = want["files"]["u"].splitlines()[1].split()
            assert [got["n_features"], got["n_cells"], got["total"][0]] == [int(x) for x in hdr]


def test_abi_findings(ctx):
    rng = np.random.default_rng(77)
    bam, stream = bamgen.tagged_bam(rng, n_cells=8, genes=20, sort_cells=False)
    got = ctx.umi_count(stream)
    want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u"], {"in.bam": bam}.get)
    assert want["exit"] == 1 and got["code"] == 17  # FQG_E_UMI_NOT_SORTED
    bam, stream = bamgen.tagged_bam(rng, n_cells=3, genes=5, gene_prefix=b"ENSG0000000000000000000")
    assert ctx.umi_count(stream)["code"] == 18      # FQG_E_UMI_FEATURE_NAME
    bam, stream = bamgen.tagged_bam(rng, n_cells=8, genes=20, sort_cells=False)
    got = ctx.umi_count(stream, sorted_by_cell=False, max_cells=5)
    want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--not_sorted_by_cell", "--max_cells", "5"],
                                {"in.bam": bam}.get)
    assert want["exit"] == 1 and got["code"] == 20 and ("Too many cells %d " % got["aux"]) in want["stderr"]


def test_fixed_geometry_batch_against_the_tree_replay(ctx):
    """BASELINE.json configs[3] geometry at 1/10 size, answer as the reference gives it: the oracle's RL_Tree
    replay over the generated (cell, gene, UMI) columns + numpy for the counters and the output rules."""
    rng = np.random.default_rng(2024)
    rec, cell, gene, umi = bamgen.config4(rng, n_cells=1500, n_genes=20000, n_triples=500000)
    stream = bamgen.header() + rec.tobytes()
    got = ctx.umi_count(stream)
    lu, lr, tu, tr, n_cells, n_genes, stats = bamgen.expected_matrix_reference(cell, gene, umi)
    assert got["code"] == 0 and (got["n_cells"], got["n_features"]) == (n_cells, n_genes)
    assert got["rl_unresolved"] == 0
    assert got["entries"][0] == lu and got["entries"][1] == lr
    assert got["total"] == [tu, tr]
    assert got["rl_replayed"] >= 1 and stats[2] >= 1   # the input does exercise the defect


@pytest.mark.parametrize("seed", range(6))
def test_dense_umi_reuse_against_the_tree_replay(ctx, seed):
    """Few UMI values, many reads per (cell, gene): most sets hit the overwrite, many read slots that only an
    earlier cell of the same gene wrote (the look-back), some read memory the reference never wrote."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(2000, 60000))
    n_cells, n_genes = int(rng.integers(2, 80)), int(rng.integers(1, 30))
    cell = np.sort(rng.integers(0, n_cells, n))
    gene = rng.zipf(1.5, n) % n_genes
    space = int(rng.choice([300, 5000, 4 ** 10]))
    umi = (rng.integers(0, space, n) * int(rng.choice([1, 7, 64]))) % (4 ** 10)
    cells_code = rng.choice(np.uint64(1) << np.uint64(32), size=n_cells, replace=False).astype(np.uint64)
    rec = bamgen.fixed_records(cells_code[cell], gene, umi.astype(np.uint64))
    got = ctx.umi_count(bamgen.header() + rec.tobytes())
    lu, lr, tu, tr, nc, ng, stats = bamgen.expected_matrix_reference(cell, gene, umi)
    assert got["code"] == 0 and got["rl_unresolved"] == 0
    assert got["entries"][0] == lu and got["entries"][1] == lr and got["total"] == [tu, tr]


def test_strict_set_extra_against_numpy(ctx):
    """strict_set = 1 (an extra, off by default): the set that src/range_list.h:150-162 documents; the matrix numpy
    derives from the generated columns (distinct counts, first-appearance ids, early break)."""
    rng = np.random.default_rng(2024)
    rec, cell, gene, umi = bamgen.config4(rng, n_cells=1500, n_genes=20000, n_triples=500000)
    stream = bamgen.header() + rec.tobytes()
    got = ctx.umi_count(stream, strict_set=True)
    c, g, u, r, n_cells, n_genes = bamgen.expected_matrix(cell, gene, umi)
    assert got["code"] == 0 and (got["n_cells"], got["n_features"]) == (n_cells, n_genes)
    assert got["entries"][0] == list(zip(g.tolist(), c.tolist(), u.tolist()))
    assert got["entries"][1] == list(zip(g.tolist(), c.tolist(), r.tolist()))
    assert got["total"] == [int(u.sum()), int(r.sum())]
    assert got["tot_reads"] == float(len(cell)) and got["n_alignments"] == len(cell)


@pytest.mark.parametrize("fresh", [True, False])
def test_reference_binary_differential(fresh):
    """The reference program itself (oracle/_ref, built from the reference's sources) on a CR-sorted
    BAM of 300 k alignments - with UMI ids that only grow (its RL_Tree is a set there) and with UMIs
    re-used across genes and cells (it is not): all three output files of both programs must be
    byte-identical."""
    ref = os.path.join(REPO, "oracle", "_ref", "bam_umi_count")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/bam_umi_count not built")
    rng = np.random.default_rng(31)
    rec, *_ = bamgen.config4(rng, n_cells=800, n_genes=5000, n_triples=230000, fresh_umis=fresh)
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "in.bam"), "wb") as f:
            f.write(bamgen.bgzf(bamgen.header() + rec.tobytes(), level=1))
        outs = {}
        for tag, exe in (("ref", ref), ("gpu", BIN)):
            p = subprocess.run(["bam_umi_count", "--bam", "in.bam", "--ucounts", tag + "_u", "--rcounts", tag + "_r"],
                               executable=exe, cwd=tmp, capture_output=True, timeout=600)
            assert p.returncode == 0, p.stderr[-300:]
            outs[tag] = p.stderr.decode("latin-1").replace(tag + "_", "X_")
            for base in ("_u", "_r"):
                for ext in ("", "_rows", "_cols"):
                    outs[tag + base + ext] = open(os.path.join(tmp, tag + base + ext), "rb").read()
        assert outs["ref"] == outs["gpu"]
        for base in ("_u", "_r"):
            for ext in ("", "_rows", "_cols"):
                assert outs["ref" + base + ext] == outs["gpu" + base + ext], base + ext


@pytest.mark.parametrize("order", [[1, 2, 3, 1], [1, 2, 1, 3], [1, 1, 2, 3, 4, 2, 5], [1, 2, 3, 4, 5, 6, 7, 8, 3], [1, 2, 2, 1]],
                         ids=lambda o: "".join(map(str, o)))
@pytest.mark.parametrize("extra", [[], ["--min_reads", "2"]], ids=["plain", "min_reads"])
def test_a_bam_that_is_not_grouped_by_cell_leaves_the_files_the_reference_leaves(order, extra):
    """src/bam_umi_count.c:1000-1015: a cell's lines are written when the cell changes, and the alignment that is out of
    order is met before the cell in front of it is written - the files hold every complete cell but that one, behind a
    header that is never finished.  Status, stderr and both matrix files as the reference binary leaves them."""
    ref = os.path.join(REPO, "oracle", "_ref", "bam_umi_count")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/bam_umi_count not built")
    rng = np.random.default_rng(sum(order) * 7 + len(order))
    bam = bamgen.bgzf(bamgen.cells_in_runs(rng, order), level=1)
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "in.bam"), "wb") as f:
            f.write(bam)
        outs = {}
        for tag, exe in (("ref", ref), ("gpu", BIN)):
            p = subprocess.run(["bam_umi_count", "--bam", "in.bam", "--ucounts", tag + "_u", "--rcounts", tag + "_r"] + extra,
                               executable=exe, cwd=tmp, capture_output=True, timeout=600)
            files = {}
            for base in ("_u", "_r"):
                for ext in ("", "_rows", "_cols"):
                    path = os.path.join(tmp, tag + base + ext)
                    files[base + ext] = open(path, "rb").read() if os.path.exists(path) else None
            outs[tag] = (p.returncode, p.stderr.decode("latin-1").replace(tag + "_", "X_"), files)
        assert outs["ref"][0] == 1 and "does not seem to be sorted by CR" in outs["ref"][1]
        assert outs["gpu"][0] == outs["ref"][0]
        assert outs["gpu"][1] == outs["ref"][1]
        assert outs["gpu"][2] == outs["ref"][2]


def split_at_cell_boundaries(stream, n_shards):
    """shards of an inflated CR-sorted BAM stream, each a BAM stream of its own holding whole cells"""
    hdr_len = None
    recs, cells = [], []
    raw = stream
    # header length: the generator's header
    hdr = bamgen.header()
    assert raw.startswith(hdr)
    p = len(hdr)
    import struct
    while p < len(raw):
        (block,) = struct.unpack_from("<i", raw, p)
        recs.append(raw[p:p + 4 + block])
        p += 4 + block
    for tid, flag, aux in uo.bam_records(stream):
        cells.append(uo.get_tag(aux, b"CR"))
    bounds = [0]
    for i in range(1, len(recs)):
        if cells[i] != cells[i - 1]:
            bounds.append(i)
    bounds.append(len(recs))
    n_cells = len(bounds) - 1
    cuts = [bounds[(n_cells * k) // n_shards] for k in range(n_shards)] + [len(recs)]
    return [hdr + b"".join(recs[cuts[k]:cuts[k + 1]]) for k in range(n_shards)]


@pytest.mark.parametrize("n_shards", [1, 2, 3, 7])
def test_shards_at_cell_boundaries_give_the_file(ctx, n_shards):
    """bam_umi_count over several GPUs: shards count on their own, agree on the global
    first-appearance ids, and apply the output rules with them.  The concatenated lines of the
    shards must be the single-GPU result = the oracle's file."""
    from fastq_utils_amd import dist as fdist

    rng = np.random.default_rng(100 + n_shards)
    # (fresh UMIs: a replayed UMI set looks back into earlier cells of its gene, which a shard does not hold)
    bam, stream = bamgen.tagged_bam(rng, n_cells=40, genes=150, reads_per_cell=(1, 200), umi_len=8, fresh_umis=True)
    want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"], {"in.bam": bam}.get)
    assert want["exit"] == 0
    whole = ctx.umi_count(stream)
    assert whole["entries"][0] == lines_of(want["files"]["u"])
    infos = []
    shards = split_at_cell_boundaries(stream, n_shards)
    for s in shards:
        infos.append(ctx.umi_count(s, defer_output=True))
    m = fdist.merge_umi_shards(infos)
    assert m["finding"] is None
    assert [f.decode() for f in m["features"]] == [ln.split("\t")[1] for ln in want["files"]["u_rows"].splitlines()]
    assert [uo.uint_642char(c).decode() for c in m["cells"]] == [ln.split("\t")[1] for ln in want["files"]["u_cols"].splitlines()]
    lines_u, lines_r, tot = [], [], [0, 0]
    for k, s in enumerate(shards):
        ctx.umi_count(s, defer_output=True)          # (one context: count again, then emit with the global ids)
        e = ctx.umi_emit(m["remap"][k], m["cell_offset"][k])
        lines_u += e["entries"][0]
        lines_r += e["entries"][1]
        tot = [tot[0] + e["total"][0], tot[1] + e["total"][1]]
    assert lines_u == whole["entries"][0] and lines_r == whole["entries"][1]
    assert tot == whole["total"]
    assert fdist.unit_float(sum(i["n_counted"] for i in infos)) == whole["tot_reads"]
    assert fdist.unit_float(sum(i["n_new"] for i in infos)) == whole["tot_umi"]


@pytest.mark.parametrize("n_shards", [2, 3, 7])
@pytest.mark.parametrize("seed", range(4))
def test_shards_reproduce_the_tree_state_across_cuts(ctx, n_shards, seed):
    """Few UMI values, many reads per (cell, gene): most sets hit the RL_Tree's overwrite and read slots that only an
    EARLIER cell of the same gene wrote - cells that lie on another shard once the file is cut.  The sharded protocol
    (dist.umi_count_sharded's steps, run here as virtual ranks on one GPU) must give the reference's file: every shard
    gets the earlier alignments of the replayed features in front of its own (src/bam_umi_count.c:418-441 keeps the
    tree arrays from cell to cell, src/range_list.c:187-198 clears only the root)."""
    from fastq_utils_amd import dist as fdist

    rng = np.random.default_rng(7000 + 10 * seed + n_shards)
    n = int(rng.integers(4000, 40000))
    n_cells, n_genes = int(rng.integers(2 * n_shards, 90)), int(rng.integers(1, 25))
    cell = np.sort(rng.integers(0, n_cells, n))
    gene = rng.zipf(1.5, n) % n_genes
    space = int(rng.choice([300, 5000]))
    umi = (rng.integers(0, space, n) * int(rng.choice([1, 7, 64]))) % (4 ** 10)
    cells_code = rng.choice(np.uint64(1) << np.uint64(32), size=n_cells, replace=False).astype(np.uint64)
    rec = bamgen.fixed_records(cells_code[cell], gene, umi.astype(np.uint64))
    hdr = bamgen.header()
    lu, lr, tu, tr, nc, ng, stats = bamgen.expected_matrix_reference(cell, gene, umi)
    whole = ctx.umi_count(hdr + rec.tobytes())
    assert whole["code"] == 0 and whole["entries"][0] == lu and whole["entries"][1] == lr and whole["rl_replayed"] >= 1
    # cut at cell boundaries
    starts = np.nonzero(np.diff(cell, prepend=-1))[0]
    cuts = [int(starts[(len(starts) * k) // n_shards]) for k in range(n_shards)] + [n]
    shards = [hdr + rec[cuts[k]:cuts[k + 1]].tobytes() for k in range(n_shards)]
    infos, umis = [], []
    for s_ in shards:
        i = ctx.umi_count(s_, defer_output=True)
        assert i["code"] == 0
        infos.append(i)
        umis.append(ctx.umi_umis())
    m = fdist.merge_umi_shards(infos)
    assert m["finding"] is None
    table = fdist.umi_global_table(umis)          # the file's UMI numbers: the tree holds numbers, not barcodes
    carried = set()
    for s_ in shards:
        i = ctx.umi_count(s_, defer_output=True, umi_table=table)
        carried |= set(fdist.umi_replayed_names(ctx, i))
    carried = sorted(carried)
    assert carried
    blobs = []
    for s_ in shards:
        i = ctx.umi_count(s_, defer_output=True, umi_table=table)
        blobs.append(fdist.umi_records_of(ctx, s_, i, carried))
    gid = {name: k + 1 for k, name in enumerate(m["features"])}
    got_u, got_r, tot, n_new, n_counted, undefined, with_history = [], [], [0, 0], 0, 0, 0, 0
    for k, s_ in enumerate(shards):
        counted, n_hist, hist = fdist.umi_count_behind(ctx, s_, b"".join(blobs[:k]), umi_table=table)
        with_history += 1 if (n_hist and counted["rl_replayed"]) else 0
        mine = fdist.umi_finish_shard(ctx, counted, n_hist, hist, gid, m["cell_offset"][k])
        got_u += mine["entries"][0]
        got_r += mine["entries"][1]
        tot = [tot[0] + mine["total"][0], tot[1] + mine["total"][1]]
        n_new += mine["n_new"]
        n_counted += mine["n_counted"]
        undefined += mine["rl_undefined"]
    assert with_history >= 1                      # the cuts do separate replayed sets from their history
    assert got_u == lu and got_r == lr and tot == [tu, tr]
    assert undefined == whole["rl_undefined"]     # (what the whole file reads of never-written memory, no more)
    assert fdist.unit_float(n_counted) == whole["tot_reads"] and fdist.unit_float(n_new) == whole["tot_umi"]


@pytest.mark.parametrize("n_shards", [2, 5])
@pytest.mark.parametrize("seed", range(3))
def test_shards_with_fractional_increments_continue_the_float_chain(ctx, n_shards, seed):
    """NH > 1 and several genes per alignment: increments are fractions, and db->tot_reads_obs / tot_umi_obs are ONE
    float32 chain over the file (src/bam_umi_count.c:490-507).  Shards count rank after rank, each starting from the
    totals of the one before (db_start / db_skip); everything else as in the unit case."""
    from fastq_utils_amd import dist as fdist

    rng = np.random.default_rng(8100 + 10 * seed + n_shards)
    bam, stream = bamgen.tagged_bam(rng, n_cells=30, genes=25, reads_per_cell=(20, 300), umi_len=4, nh=True, multi_gx=True)
    want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"], {"in.bam": bam}.get)
    assert want["exit"] == 0
    whole = ctx.umi_count(stream)
    assert whole["code"] == 0 and not whole["unit_increments"]
    assert whole["entries"][0] == lines_of(want["files"]["u"]) and whole["entries"][1] == lines_of(want["files"]["r"])
    shards = split_at_cell_boundaries(stream, n_shards)
    infos, umis = [], []
    for s_ in shards:
        infos.append(ctx.umi_count(s_, defer_output=True))
        umis.append(ctx.umi_umis())
    m = fdist.merge_umi_shards(infos)
    assert m["finding"] is None
    table = fdist.umi_global_table(umis)
    carried, numbered = set(), []
    for s_ in shards:
        i = ctx.umi_count(s_, defer_output=True, umi_table=table)
        carried |= set(fdist.umi_replayed_names(ctx, i))
    blobs = []
    for s_ in shards:
        i = ctx.umi_count(s_, defer_output=True, umi_table=table)
        blobs.append(fdist.umi_records_of(ctx, s_, i, sorted(carried)) if carried else b"")
    gid = {name: k + 1 for k, name in enumerate(m["features"])}
    got_u, got_r, chain = [], [], (0.0, 0.0)
    for k, s_ in enumerate(shards):
        hdr, _, used = fdist.bam_split(s_)
        history = b"".join(blobs[:k])
        hist, n_hist = {"n_new": 0, "n_counted": 0}, 0
        if history:
            hist = ctx.umi_count(hdr + history, defer_output=True, umi_table=table)
            n_hist = len(hist["cells"])
        counted = ctx.umi_count(hdr + history + s_[len(hdr):used], defer_output=True, umi_table=table, db_start=chain,
                                db_skip=hist.get("n_alignments", 0) if history else 0)
        chain = (counted["tot_reads"], counted["tot_umi"])
        mine = fdist.umi_finish_shard(ctx, counted, n_hist, hist, gid, m["cell_offset"][k])
        got_u += mine["entries"][0]
        got_r += mine["entries"][1]
    assert got_u == whole["entries"][0] and got_r == whole["entries"][1]
    assert chain == (whole["tot_reads"], whole["tot_umi"])


def test_sharded_protocol_through_a_one_rank_group(ctx):
    import torch
    import torch.distributed as dist
    from fastq_utils_amd import dist as fdist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(5)
        bam, stream = bamgen.tagged_bam(rng, n_cells=25, genes=80, fresh_umis=True)
        whole = ctx.umi_count(stream)
        got = fdist.umi_count_sharded(ctx, stream)
        assert got["finding"] is None and got["entries"] == whole["entries"]
        assert (got["n_entries"], got["total"], got["tot_reads"], got["tot_umi"]) == (
            whole["n_entries"], whole["total"], whole["tot_reads"], whole["tot_umi"])
    finally:
        dist.destroy_process_group()


def test_record_index_resident_on_the_device(ctx):
    """FQG_MEM_DEVICE_INDEXED: records and their offsets in device memory give what host offsets give"""
    import numpy as np

    torch = pytest.importorskip("torch")
    from tests import bamgen

    rng = np.random.default_rng(12)
    rec, _, _, _ = bamgen.config4(rng, n_cells=40, n_genes=300, n_triples=20000)
    hdr = bamgen.header()
    n = rec.shape[0]
    blob = hdr + rec.tobytes()
    dev = torch.device("cuda", 0)
    stream = torch.frombuffer(bytearray(blob + b"\0" * 64), dtype=torch.uint8).to(dev)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(bamgen.REC_BYTES) + np.uint64(len(hdr))
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    a = ctx.umi_count(blob)
    b = ctx.umi_count(stream.data_ptr(), offsets=list(int(x) for x in offs), nbytes=len(blob))
    c = ctx.umi_count(stream.data_ptr(), nbytes=len(blob), offsets_device=(d_offs.data_ptr(), n))
    assert a["code"] == b["code"] == c["code"] == 0
    for k in ("entries", "features", "cells", "n_counted", "n_new", "rl_replayed"):
        assert a[k] == b[k] == c[k], k


# ---- bin/bam_umi_count over several contexts (FQGPU_DEVICES; host/umi_multi.h: the cell-range protocol of
# dist.umi_count_sharded inside the drop-in program) ---------------------------------------------------------------
def multi_case_run(i):
    case = CLI_CASES[i]
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        real = [a.replace("OUTU", rel + "/u.mtx").replace("OUTR", rel + "/r.mtx") for a in case["args"]]
        p = subprocess.run(["bam_umi_count"] + real, executable=BIN, cwd=GOLD, capture_output=True, timeout=300,
                           env=dict(os.environ, FQGPU_DEVICES="0,0,0"))
        got = {}
        for base in ("u.mtx", "r.mtx"):
            for ext in ("", "_rows", "_cols"):
                path = os.path.join(tmp, base + ext)
                if os.path.exists(path):
                    got["SCRATCH/" + base + ext] = open(path, "rb").read().decode("latin-1")
    return p.returncode, p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), got


MULTI_RUNS = SideBySide(multi_case_run, range(len(CLI_CASES)))


@pytest.mark.parametrize("i", range(len(CLI_CASES)), ids=[" ".join(c["args"])[:80] for c in CLI_CASES])
def test_cli_over_three_contexts_matches_reference_binary(i):
    """every golden invocation once more with FQGPU_DEVICES=0,0,0: the files of the reference binary, byte for byte -
    whether the cells went over the contexts or the program fell back to one (findings, unsorted input, one cell)"""
    case = CLI_CASES[i]
    rc, err, got = MULTI_RUNS.get(i)
    assert (rc if rc >= 0 else 128 - rc) == want_exit(case), err[-400:]
    if case["exit"] == -6:
        assert "Assertion `len1+1 < FEAT_ID_MAX_LEN' failed" in err
        return
    assert err == case["stderr"]
    if case["exit"] == 0:
        assert got == golden_files(case)


@pytest.mark.parametrize("variant", ["plain", "nh", "multi", "noise", "whitelist", "thresholds", "dense"])
def test_program_over_contexts_against_the_oracle(variant):
    """seeded CR-sorted BAMs with re-used UMIs (the RL_Tree's defects: rounds 2 and 3 of the protocol), NH weights and
    several genes per alignment (the float32 chain of totals, shard after shard): the program with 2 and 3 contexts
    writes the oracle's files and says the oracle's stderr - and it did go over the contexts"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("m" + variant).encode()) % 9999)  # (the same files in every run: not hash())
    for trial in range(2):
        kw = dict(n_cells=int(rng.integers(8, 80)), genes=int(rng.integers(5, 300)), umi_len=int(rng.integers(2, 9)),
                  reads_per_cell=(1, int(rng.integers(2, 300))))
        if variant == "dense":   # few genes, few UMIs, many reads: nearly every set is replayed as the tree behaves
            kw = dict(n_cells=int(rng.integers(20, 60)), genes=3, umi_len=3, reads_per_cell=(100, 400))
        if variant in ("nh", "multi", "noise"):
            kw["nh"] = True
        if variant in ("multi", "noise"):
            kw["multi_gx"] = True
        if variant == "noise":
            kw["noise"] = True
        bam, stream = bamgen.tagged_bam(rng, **kw)
        extra = []
        files = {"in.bam": bam}
        if variant == "thresholds":
            extra += ["--min_reads", "2", "--min_umis", "2"]
        if variant == "whitelist":
            cells = []
            for tid, flag, aux in uo.bam_records(stream):
                c = uo.get_tag(aux, b"CR")
                if c and c not in cells:
                    cells.append(c)
            files["wl.txt"] = b"".join(c + b"\n" for c in cells[::2])
            extra += ["--known_cells", "wl.txt"]
        want = uo.run_bam_umi_count(["--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"] + extra, files.get)
        for devs in ("0,0", "0,0,0"):
            with tempfile.TemporaryDirectory() as tmp:
                for name, data in files.items():
                    with open(os.path.join(tmp, name), "wb") as f:
                        f.write(data)
                p = subprocess.run(["bam_umi_count", "--bam", "in.bam", "--ucounts", "u", "--rcounts", "r"] + extra, executable=BIN,
                                   cwd=tmp, capture_output=True, timeout=300,
                                   env=dict(os.environ, FQGPU_DEVICES=devs, FQGPU_UMI_MULTI_DEBUG="1"))
                err = p.stderr.decode("latin-1")
                went = "[umi multi] counted over the devices" in err
                err = "".join(ln for ln in err.splitlines(True) if not ln.startswith("[umi multi]"))
                assert p.returncode == want["exit"], err[-500:]
                assert err == want["stderr"], (variant, devs)
                if want["exit"] == 0:
                    assert went, (variant, devs, p.stderr.decode("latin-1")[-300:])
                    for name, text in want["files"].items():
                        if name in files:   # (the inputs)
                            continue
                        have = open(os.path.join(tmp, name), "rb").read()
                        assert (have if isinstance(text, bytes) else have.decode("latin-1")) == text, (variant, devs, name)
