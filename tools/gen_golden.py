#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container (needs /root/reference and oracle/_ref/, see oracle/Makefile):

    make -C oracle ref && python tools/gen_golden.py

What it writes
  tests/golden/data/*.fastq.gz    the reference's own FASTQ test fixtures (data files, copied
                                  byte for byte from <reference>/tests/)
  tests/golden/fastq_info.json    one entry per fastq_info invocation: argv, exit status, stdout,
                                  stderr, all captured from oracle/_ref/fastq_info (the reference
                                  program compiled unmodified)

The invocation list is the fastq_info section of the reference's run_tests.sh (:252-343) plus a
sweep of every fixture through the single-file modes, and every _1/_2 pair through the paired
modes.  Nothing here is needed at test time on the GPU box; only the outputs are.
"""
import glob
import gzip
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FQ_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(GOLD, "data")
REF_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_info")


def run(args):
    p = subprocess.run([REF_BIN] + args, cwd=GOLD, capture_output=True, timeout=120)
    return {
        "args": args,
        "exit": p.returncode,
        "stdout": p.stdout.decode("latin-1"),
        "stderr": p.stderr.decode("latin-1"),
    }


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(DATA, exist_ok=True)
    names = []
    for src in sorted(glob.glob(os.path.join(REF, "tests", "*.fastq.gz"))):
        dst = os.path.join(DATA, os.path.basename(src))
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        names.append(os.path.basename(src))
    # an empty plain file, as run_tests.sh:298 creates with touch
    open(os.path.join(DATA, "empty.fastq"), "wb").close()

    d = lambda n: "data/" + n
    jobs = []
    singles = names + ["empty.fastq"]
    for n in singles:
        for flags in ([], ["-r"], ["-q"], ["-e"], ["-r", "-q"], ["-r", "-e"]):
            jobs.append(flags + [d(n)])
        jobs.append([d(n), "pe"])
    pairs = []
    for n in names:
        if "_1." in n:
            m = n.replace("_1.", "_2.")
            if m in names:
                pairs.append((n, m))
    # cross pairs exercised by run_tests.sh
    pairs += [
        ("test_e19_1.fastq.gz", "test_empty.fastq.gz"),
        ("test_empty.fastq.gz", "test_e19_1.fastq.gz"),
        ("test_empty.fastq.gz", "test_1.fastq.gz"),
        ("test_1.fastq.gz", "test_empty.fastq.gz"),
        ("pe_bug14.fastq.gz", "pe_bug14.fastq.gz"),
        ("casava.1.8_readname_trunc_1.err.fastq.gz", "casava.1.8_readname_trunc_2.fastq.gz"),
        ("casava.1.8_readname_trunc_2.fastq.gz", "casava.1.8_readname_trunc_1.err.fastq.gz"),
        ("casava.1.8_readname_trunc_1.err2.fastq.gz", "casava.1.8_readname_trunc_2.fastq.gz"),
        ("casava.1.8_readname_trunc_1.fastq.gz", "casava.1.8_2.fastq.gz"),
        ("test_1.fastq.gz", "test_2.fastq.gz"),
        ("test_2.fastq.gz", "test_1.fastq.gz"),
        ("test_21_1.fastq.gz", "test_1.fastq.gz"),
    ]
    for a, b in pairs:
        for x, y in ((a, b), (b, a)):
            for flags in ([], ["-r", "-s"], ["-s"], ["-r"], ["-q"]):
                jobs.append(flags + [d(x), d(y)])
    seen, out = set(), []
    for j in jobs:
        k = tuple(j)
        if k in seen:
            continue
        seen.add(k)
        out.append(run(j))
    with open(os.path.join(GOLD, "fastq_info.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("fastq_info invocations:", len(out), "by exit status:", by)


PB_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_pre_barcodes")


def pre_barcodes_jobs():
    """argv lists for fastq_pre_barcodes; OUT1/OUT2 stand for output files in a scratch directory.
    Taken from run_tests.sh:383-449 and sh/fastq2bam:116-273 (the 10x / drop-seq flag sets)."""
    d = "data/"
    b1, b2 = d + "barcode_test2_1.fastq.gz", d + "barcode_test2_2.fastq.gz"
    common = ["--phred_encoding", "33", "--umi_read", "index1", "--umi_offset", "0", "--umi_size", "16",
              "--read1_offset", "0", "--read1_size", "-1"]
    jobs = []
    jobs.append(["--index1", d + "barcode_test_1.fastq.gz", "--min_qual", "10"] + common +
                ["--read1", d + "barcode_test_2.fastq.gz", "--outfile1", "OUT1"])
    jobs.append(jobs[-1] + ["-X"])
    jobs.append(["--index1", b1, "--min_qual", "10"] + common + ["--read1", b2, "--outfile1", "OUT1"])  # pre1
    cell = ["--cell_read", "index1", "--cell_offset", "0", "--cell_size", "8"]
    jobs.append(["--index1", b1, "--min_qual", "10"] + common + cell + ["--read1", b2, "--outfile1", "OUT1"])  # pre2
    samp = ["--sample_read", "read1", "--sample_offset", "0", "--sample_size", "4"]
    jobs.append(["--index1", b1, "--min_qual", "1"] + common + cell + samp + ["--read1", b2, "--outfile1", "OUT1"])  # pre3
    three = ["--index1", b1, "--index2", b1, "--index3", b1, "--min_qual", "1"] + common + [
        "--cell_read", "index2", "--cell_offset", "0", "--cell_size", "8", "--sample_read", "index3",
        "--sample_offset", "0", "--sample_size", "4", "--read1", b2]
    jobs.append(three + ["--outfile1", "OUT1"])
    jobs.append(three + ["--read2", b2, "--outfile1", "OUT1", "--outfile2", "OUT2"])
    jobs.append(three + ["--read2", b2, "--outfile1", "OUT1", "--outfile2", "OUT2", "--sam"])
    jobs.append(three + ["--outfile1", "OUT1", "--sam"])
    jobs.append(["--index1", b1, "--min_qual", "10"] + common + cell + ["--read1", b2, "--outfile1", "OUT1", "--sam"])
    jobs.append(["--index1", b1, "--min_qual", "10"] + common + cell + ["--read1", b2, "--outfile1", "OUT1", "--sam", "--10x"])
    umi16 = ["--umi_read", "index1", "--umi_offset", "0", "--umi_size", "16"]
    inter, cas = d + "inter.fastq.gz", d + "casava.1.8i.fastq.gz"
    jobs.append(["--interleaved", "read1", "--read1", inter, "--index1", inter, "--outfile1", "OUT1"] + umi16 + ["--sam"])
    jobs.append(["--interleaved", "read1,read2,index1", "--read1", inter, "--index1", inter, "--outfile1", "OUT1"] + umi16 + ["--sam"])
    for order in ("index1,read1", "read1,index1"):
        for f in (cas, inter):
            jobs.append(["--interleaved", order, "--read1", f, "--index1", f, "--outfile1", "OUT1"] + umi16)
            jobs.append(["--interleaved", order, "--read1", f, "--index1", f, "--outfile1", "OUT1"] + umi16 + ["--sam"])
    # slicing variants
    for off, size in (("0", "10"), ("5", "10"), ("5", "-1"), ("0", "0"), ("3", "200"), ("0", "98"), ("0", "97"), ("1", "97")):
        jobs.append(["--index1", b1, "--min_qual", "1", "--phred_encoding", "33"] + umi16 + cell +
                    ["--read1_offset", off, "--read1_size", size, "--read1", b2, "--outfile1", "OUT1"])
        if (off, size) != ("5", "-1"):  # the reference itself crashes (SIGSEGV) on this one in SAM mode
            jobs.append(jobs[-1] + ["--sam"])
    # 10x v2 flag set (sh/fastq2bam:125-130) on the pbmc8k fixtures, and the tx (10x v1) set
    r1, r2 = d + "pbmc8k_S1_L007_R1_001.fastq.gz", d + "pbmc8k_S1_L007_R2_001.fastq.gz"
    v2 = ["--read1", r2, "--index1", r1, "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
          "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16"]
    jobs.append(v2 + ["--outfile1", "OUT1"])
    jobs.append(v2 + ["--outfile1", "OUT1", "--sam"])
    jobs.append(v2 + ["--outfile1", "OUT1", "--sam", "--10x", "--phred_encoding", "33", "--min_qual", "10"])
    jobs.append(v2 + ["--outfile1", "OUT1", "--phred_encoding", "33", "--min_qual", "30"])
    tx = ["--read1", d + "tx.RA.fastq.gz", "--index3", d + "tx.RA.fastq.gz", "--index1", d + "tx.I1.fastq.gz",
          "--umi_read", "index3", "--umi_offset", "0", "--umi_size", "10", "--cell_read", "index1",
          "--cell_offset", "0", "--cell_size", "14", "--interleaved", "read1,index3",
          "--index2", d + "tx.I2.fastq.gz", "--sample_read", "index2", "--sample_offset", "0", "--sample_size", "8"]
    jobs.append(tx + ["--sam", "--outfile1", "-"])
    jobs.append(tx + ["--outfile1", "OUT1"])
    # failures
    jobs.append([])
    jobs.append(["--index1", d + "test_1.fastq.gz", "--min_qual", "1"] + common + cell + samp + ["--read1", b2, "--outfile1", "OUT1"])
    jobs.append(["--index1", b1, "--min_qual", "1"] + common + cell + samp + ["--read1", b2])
    jobs.append(["--index1", b1, "--min_qual", "1"] + common + ["--umi_read", "index5", "--read1", b2, "--outfile1", "OUT1"])
    jobs.append(["--index1", b1, "--min_qual", "1"] + common + ["--read1", d + "test_e5.fastq.gz", "--outfile1", "OUT1"])
    jobs.append(["--help"])
    return jobs


def gen_pre_barcodes():
    import gzip
    import tempfile

    if not os.path.exists(PB_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    out = []
    for args in pre_barcodes_jobs():
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = [a.replace("OUT1", rel + "/o1.fastq.gz").replace("OUT2", rel + "/o2.fastq.gz") for a in args]
            p = subprocess.run(["fastq_pre_barcodes"] + real, executable=PB_BIN, cwd=GOLD, capture_output=True, timeout=120)
            files = {}
            for tag, fn in (("OUT1", "o1.fastq.gz"), ("OUT2", "o2.fastq.gz")):
                path = os.path.join(tmp, fn)
                if os.path.exists(path):
                    raw = open(path, "rb").read()
                    try:
                        files[tag] = gzip.decompress(raw).decode("latin-1") if raw else ""
                    except Exception:
                        files[tag] = None  # unfinished gzip stream (the reference exited mid-way)
            out.append({"args": args, "exit": p.returncode,
                        "stdout": p.stdout.decode("latin-1").replace(rel + "/", "SCRATCH/"),
                        "stderr": p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), "files": files})
    with open(os.path.join(GOLD, "pre_barcodes.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("fastq_pre_barcodes invocations:", len(out), "by exit status:", by)


UMI_BIN = os.path.join(REPO, "oracle", "_ref", "bam_umi_count")
UMI_DATA = os.path.join(GOLD, "data_umi")


def umi_synthetic_bams():
    """Seeded BAMs from tests/bamgen.py (committed under tests/golden/data_umi/ so that the GPU box
    needs no generator run): name -> bytes"""
    import numpy as np

    sys.path.insert(0, REPO)
    from tests import bamgen

    out = {}
    rng = np.random.default_rng(20251002)
    # fresh_umis: UMI ids only grow, the regime in which the reference's RL_Tree is a set
    out["syn_plain.bam"] = bamgen.tagged_bam(rng, n_cells=30, genes=60, fresh_umis=True)[0]
    out["syn_nh.bam"] = bamgen.tagged_bam(rng, n_cells=20, genes=50, nh=True, fresh_umis=True)[0]
    out["syn_multi.bam"] = bamgen.tagged_bam(rng, n_cells=20, genes=50, nh=True, multi_gx=True, noise=True,
                                             fresh_umis=True)[0]
    out["syn_few_umis.bam"] = bamgen.tagged_bam(rng, n_cells=25, genes=300, reads_per_cell=(1, 6), noise=True,
                                                fresh_umis=True)[0]
    # UMIs re-used across cells and genes, as in real data: the reference's RL_Tree loses / invents
    # members here (test_oracle_umi.py::test_reference_rl_tree_defect)
    out["syn_reuse.bam"] = bamgen.tagged_bam(rng, n_cells=30, genes=60)[0]
    # the smallest input that shows it: one cell; 40 UMIs on gene A (ids 1..40); gene B sees UMI 40,
    # UMI 20, UMI 40 again.  A set holds {20, 40} for B; the reference counts 3 (inserting 20 drops 40:
    # shift_right() in src/range_list.c:287-301 moves nothing when one node follows the new one)
    umis = []
    while len(umis) < 40:
        u = bamgen.barcode(rng, 10)
        if u not in umis:
            umis.append(u)
    recs = [bamgen.record(b"a%d" % i, bamgen.aux_z(b"CR", b"ACGTACGTACGT") + bamgen.aux_z(b"GX", b"A") +
                          bamgen.aux_z(b"RX", u)) for i, u in enumerate(umis)]
    for i, k in enumerate((39, 19, 39)):
        recs.append(bamgen.record(b"b%d" % i, bamgen.aux_z(b"CR", b"ACGTACGTACGT") + bamgen.aux_z(b"GX", b"B") +
                                  bamgen.aux_z(b"RX", umis[k])))
    out["rl_defect.bam"] = bamgen.bgzf(bamgen.header() + b"".join(recs))
    out["syn_unsorted.bam"] = bamgen.tagged_bam(rng, n_cells=10, genes=30, sort_cells=False, nh=True)[0]
    out["syn_long_gene.bam"] = bamgen.tagged_bam(rng, n_cells=3, genes=5, gene_prefix=b"ENSG0000000000000000000")[0]
    return out


def umi_jobs():
    """argv lists for bam_umi_count; OUTU / OUTR stand for output files in a scratch directory.  The
    first block follows the reference's own suite (run_tests.sh:96-177)."""
    d = lambda n: "data_umi/" + n
    ns = "--not_sorted_by_cell"
    jobs = [
        ["--min_reads", "1", "--bam", d("test_annot.bam"), "--ucounts", "OUTU", "-x", "TX", ns],
        ["--min_reads", "1", "--bam", d("test_annot.bam"), "--ucounts", "OUTU", "-x", "GX", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "-x", "TX", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "-x", "GX", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "-x", "TX", ns, "--10x"],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "-x", "GX", ns, "--10x"],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--multi_mapped", "--ucounts", "OUTU", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "--ignore_sample", ns,
         "--cell_suffix", "-123456789"],
        [ns, "--min_reads", "1", "--bam", d("test_annot5.bam"), "--known_cells", d("known_cells.txt"), "--ucounts", "OUTU"],
        ["--min_reads", "1"],
        ["--bam", d("test_annot.bam")],
        ["--bam", d("test_annot.bam"), "-x"],
        ["--bam", d("test_annot.bam_missing")],
        ["--bam", d("test_annot.bam_missing"), "--ucounts", "OUTU"],
        ["--min_reads", "1", "--bam", d("test_annot.bam"), "--known_umi", d("known_umis.txt_missing"), "--ucounts", "OUTU"],
        ["--min_reads", "1", "--bam", d("test_annot.bam"), "--known_cells", d("known_cells.txt_missing"), "--ucounts", "OUTU"],
        ["--sorted_by_cell", "--min_reads", "1", "--bam", d("test_annot.bam"), "--known_cells",
         d("known_cells.txt_missing"), "--ucounts", "OUTU"],
        ["--min_reads", "1", "--bam", d("test_annot.bam"), "--known_umi", d("known_umis.txt"), ns, "--ucounts", "OUTU",
         "--rcounts", "OUTR"],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--known_umi", d("known_umis.txt"), ns, "--ucounts", "OUTU"],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "-x", "TX", "--max_cells", "10",
         "--max_feat", "2", "--feat_cell", "2", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "--max_cells", "100", ns],
        ["--min_reads", "4", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "--rcounts", "OUTR", ns],
        ["--min_reads", "1", "--bam", d("test_annot5.bam"), "--ucounts", "OUTU", "--uniq_mapped", ns],
        # default (sorted-by-cell) mode
        ["--bam", d("test_one_cell.bam"), "--ucounts", "OUTU"],
        ["--bam", d("test_one_cell.bam"), "--ucounts", "OUTU", "--rcounts", "OUTR", "--min_reads", "2"],
        ["--bam", d("test_one_cell.bam"), "--ucounts", "OUTU", "--uniq_mapped", "--min_umis", "3"],
        ["--bam", d("test_one_cell.bam"), "--ucounts", "OUTU", "-x", "TX", "--cell_suffix", "-1"],
        ["--bam", d("test_annot5.bam"), "--ucounts", "OUTU"],
        ["--bam", d("test_annot.bam"), "--ucounts", "OUTU"],
        ["--help"],
    ]
    for name in ("syn_plain.bam", "syn_nh.bam", "syn_multi.bam", "syn_few_umis.bam", "syn_long_gene.bam"):
        jobs.append(["--bam", d(name), "--ucounts", "OUTU"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--rcounts", "OUTR"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--rcounts", "OUTR", "--min_reads", "2", "--min_umis", "2"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--uniq_mapped"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--10x", "-x", "TX"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", ns, "--rcounts", "OUTR"])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--known_cells", d("syn_cells.txt")])
        jobs.append(["--bam", d(name), "--ucounts", "OUTU", "--known_umi", d("known_umis.txt")])
    jobs.append(["--bam", d("rl_defect.bam"), "--ucounts", "OUTU"])
    jobs.append(["--bam", d("syn_reuse.bam"), "--ucounts", "OUTU", "--rcounts", "OUTR"])
    jobs.append(["--bam", d("syn_reuse.bam"), "--ucounts", "OUTU", "--rcounts", "OUTR", ns])
    jobs.append(["--bam", d("syn_unsorted.bam"), "--ucounts", "OUTU"])
    jobs.append(["--bam", d("syn_unsorted.bam"), "--ucounts", "OUTU", ns])
    jobs.append(["--bam", d("syn_unsorted.bam"), "--ucounts", "OUTU", ns, "--max_cells", "5"])
    jobs.append(["--bam", d("syn_plain.bam"), "--ucounts", "OUTU", "--max_feat", "20"])
    jobs.append(["--bam", d("syn_plain.bam"), "--ucounts", "OUTU", "-X", "RX"])
    return jobs


def gen_umi():
    import tempfile

    if not os.path.exists(UMI_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(UMI_DATA, exist_ok=True)
    for n in ("test_annot.bam", "test_annot5.bam", "test_one_cell.bam", "known_cells.txt", "known_umis.txt"):
        shutil.copyfile(os.path.join(REF, "tests", n), os.path.join(UMI_DATA, n))
        os.chmod(os.path.join(UMI_DATA, n), 0o644)
    syn = umi_synthetic_bams()
    for n, b in syn.items():
        with open(os.path.join(UMI_DATA, n), "wb") as f:
            f.write(b)
    # a cell whitelist that keeps about half of the cells of the synthetic BAMs
    sys.path.insert(0, REPO)
    from oracle import umi_oracle as uo
    keep = []
    for n in sorted(syn):
        seen = []
        for tid, flag, aux in uo.bam_records(uo.bgzf_inflate(syn[n])):
            c = uo.get_tag(aux, b"CR")
            if c and c not in seen:
                seen.append(c)
        keep += seen[::2]
    with open(os.path.join(UMI_DATA, "syn_cells.txt"), "wb") as f:
        f.write(b"".join(c + b"\n" for c in keep))
    out = []
    for args in umi_jobs():
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = [a.replace("OUTU", rel + "/u.mtx").replace("OUTR", rel + "/r.mtx") for a in args]
            p = subprocess.run(["bam_umi_count"] + real, executable=UMI_BIN, cwd=GOLD, capture_output=True, timeout=300)
            files = {}
            for tag, fn in (("OUTU", "u.mtx"), ("OUTR", "r.mtx")):
                for ext in ("", "_rows", "_cols"):
                    path = os.path.join(tmp, fn + ext)
                    if os.path.exists(path):
                        files[tag + ext] = open(path, "rb").read().decode("latin-1")
            out.append({"args": args, "exit": p.returncode,
                        "stdout": p.stdout.decode("latin-1").replace(rel + "/", "SCRATCH/"),
                        "stderr": p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), "files": files})
    with open(os.path.join(GOLD, "umi_count.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("bam_umi_count invocations:", len(out), "by exit status:", by)


FN_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_filter_n")
TP_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_trim_poly_at")


def filters_synthetic():
    """A seeded FASTQ with what the two filters react to: poly-A tails and poly-T heads of many lengths
    (upper / lower case, N mixed in), reads that are nothing but A or T, N-rich reads, very short reads,
    and a few records whose quality line is longer or shorter than the sequence."""
    import numpy as np

    rng = np.random.default_rng(77)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(400):
        n = int(rng.integers(1, 90))
        seq = bytearray(bases[rng.integers(0, 4, n)].tobytes())
        kind = i % 10
        if kind in (0, 1, 2):  # poly-A tail
            t = int(rng.integers(1, 25))
            tail = bytes(rng.choice(list(b"AAAAAAaN"), t).astype(np.uint8))
            seq += tail
        elif kind in (3, 4):  # poly-T head
            t = int(rng.integers(1, 25))
            seq = bytearray(bytes(rng.choice(list(b"TTTTTTtn"), t).astype(np.uint8))) + seq
        elif kind == 5:  # nothing but A / T / N
            seq = bytearray(bytes(rng.choice(list(b"AAN" if i % 20 == 5 else b"TTn"), n).astype(np.uint8)))
        elif kind == 6:  # N-rich
            for j in range(len(seq)):
                if rng.random() < 0.3:
                    seq[j] = ord("N") if rng.random() < 0.7 else ord("n")
        elif kind == 7 and n > 4:  # both ends
            seq = bytearray(b"TTTTT") + seq + bytearray(b"AAAAAA")
        qlen = len(seq)
        if i % 57 == 11:
            qlen += int(rng.integers(1, 6))  # quality longer than the sequence
        elif i % 57 == 23 and qlen > 6:
            qlen -= int(rng.integers(1, 4))  # shorter (but longer than any trimmed prefix here: kind 9 / no T head)
        qual = (rng.integers(2, 41, qlen) + 33).astype(np.uint8).tobytes()
        recs.append(b"@FLT:%d:%d/1\n" % (i % 7, i) + bytes(seq) + b"\n+\n" + qual + b"\n")
    return b"".join(recs)


def filters_jobs():
    d = lambda n: "data/" + n
    small = []
    for path in sorted(glob.glob(os.path.join(DATA, "*.fastq.gz"))):
        if os.path.getsize(path) < 60_000:
            small.append(os.path.basename(path))
    fn, tp = [], []
    # run_tests.sh:211-216
    fn += [[d("test_21_2.fastq.gz")], ["-n", "100", d("test_21_2.fastq.gz")], [d("test_1.fastq.gz")], ["--help"], []]
    fn += [["-x", d("test_1.fastq.gz")], ["-n"], ["-n", "5"], [d("does_not_exist.fastq.gz")], ["-n", "-3", d("test_1.fastq.gz")],
           ["-n", "250", d("test_1.fastq.gz")], [d("test_1.fastq.gz"), "extra"], [d("test_1.fastq.gz"), "extra", "more"]]
    for n in small + ["syn_filters.fastq", "empty.fastq", "syn_filters_trunc.fastq"]:
        for flags in ([], ["-n", "1"], ["-n", "10"], ["-n", "30"], ["-n", "100"]):
            fn.append(flags + [d(n)])
    # run_tests.sh:190-201
    tp += [[], ["--help"], ["--file", d("a_1.fastq.gz")], ["--file", d("a_1xx.fastq.gz"), "--outfile", "OUT"],
           ["--file", d("a_1.fastq.gz"), "--outfile", "/xxx/tmp.fastq.gz"],
           ["--file", d("a_1.fastq.gz"), "--outfile", "OUT", "--min_poly_at_len", "20"],
           ["--file", d("poly_at.fastq.gz"), "--outfile", "OUT", "--min_poly_at_len", "3"],
           ["--file", d("poly_at.fastq.gz"), "--outfile", "OUT", "--min_poly_at_len", "300", "--min_len", "1"],
           ["--outfile", "OUT"], ["--file", d("poly_at.fastq.gz"), "--outfile", "OUT", "--bogus", "--min_len", "5"]]
    for n in small + ["syn_filters.fastq", "empty.fastq", "syn_filters_trunc.fastq"]:
        for extra in ([], ["--min_poly_at_len", "3"], ["--min_poly_at_len", "5", "--min_len", "1"],
                      ["--min_poly_at_len", "1", "--min_len", "30"], ["--min_poly_at_len", "0"],
                      ["--min_poly_at_len", "8", "--min_len", "-1"], ["--min_poly_at_len", "2", "--min_len", "0"]):
            tp.append(["--file", d(n), "--outfile", "OUT"] + extra)
    return fn, tp


def gen_filters():
    import gzip
    import hashlib
    import tempfile

    for b in (FN_BIN, TP_BIN):
        if not os.path.exists(b):
            sys.exit("build the reference first: make -C oracle ref")
    syn = filters_synthetic()
    with open(os.path.join(DATA, "syn_filters.fastq"), "wb") as f:
        f.write(syn)
    with open(os.path.join(DATA, "syn_filters_trunc.fastq"), "wb") as f:
        f.write(syn[: syn.index(b"@FLT:5:40/1")] + b"@FLT:5:40/1\nACGT\n+\n")  # two lines short of a record
    fn, tp = filters_jobs()

    def pack(text):
        h = hashlib.sha256(text).hexdigest()
        return {"sha256": h, "len": len(text), "text": text.decode("latin-1") if len(text) <= 30000 else None}

    out = {"filter_n": [], "trim_poly_at": []}
    for args in fn:
        p = subprocess.run(["fastq_filter_n"] + args, executable=FN_BIN, cwd=GOLD, capture_output=True, timeout=120)
        out["filter_n"].append({"args": args, "exit": p.returncode, "stdout": pack(p.stdout),
                                "stderr": p.stderr.decode("latin-1")})
    for args in tp:
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = [a.replace("OUT", rel + "/o.fastq.gz") if a == "OUT" else a for a in args]
            p = subprocess.run(["fastq_trim_poly_at"] + real, executable=TP_BIN, cwd=GOLD, capture_output=True, timeout=120)
            path = os.path.join(tmp, "o.fastq.gz")
            written = None
            if os.path.exists(path) and p.returncode == 0:
                raw = open(path, "rb").read()
                written = pack(gzip.decompress(raw) if raw else b"")
            out["trim_poly_at"].append({"args": args, "exit": p.returncode, "stdout": p.stdout.decode("latin-1"),
                                        "stderr": p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"),
                                        "out": written})
    with open(os.path.join(GOLD, "filters.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    for k, v in out.items():
        by = {}
        for o in v:
            by[o["exit"]] = by.get(o["exit"], 0) + 1
        print(k, "invocations:", len(v), "by exit status:", by)


FP_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_filterpair")


def filterpair_synthetic():
    """Two seeded FASTQ files for fastq_filterpair: mates in different orders, reads without a mate in either file -
    in file 1 both before and behind the last mated read (the reference only finds the latter, src/fastq_filterpair.c:
    196-216 reads file 1 on from where its last copy ended) -, and a name that file 2 asks for twice."""
    import numpy as np

    rng = np.random.default_rng(2718)

    def rec(i, mate, n=40):
        seq = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])
        qual = bytes((rng.integers(2, 41, n) + 33).astype(np.uint8))
        return b"@SYN:1:FC:1:1:%d:%d %d:N:0:ACGT\n" % (i % 11, i, mate) + seq + b"\n+\n" + qual + b"\n"
    ids = list(range(300))
    only1 = set(int(x) for x in rng.choice(300, 40, replace=False))
    only2 = list(range(1000, 1030))
    f1 = [i for i in ids]
    f2 = [i for i in ids if i not in only1] + only2
    rng.shuffle(f2)
    f2.insert(50, f2[10])  # asked for twice: the second asker finds nothing
    return b"".join(rec(i, 1) for i in f1), b"".join(rec(i, 2) for i in f2)


def filterpair_jobs():
    d = lambda n: "data/" + n
    jobs = []
    # run_tests.sh:361-370
    jobs += [[d("test_2.fastq.gz"), d("test_2.fastq.gz")], [d("a_1.fastq.gz"), d("a_2.fastq.gz")],
             [d("casava.1.8_2.fastq.gz"), d("casava.1.8_2.fastq.gz")], [d("casava.1.8_1.fastq.gz"), d("casava.1.8_1.fastq.gz")],
             [d("c18_10000_1.fastq.gz"), d("c18_10000_2.fastq.gz")], [d("c18_10000_1.fastq.gz"), d("c18_10000_2.fastq.gz"), "sorted"],
             [d("c18_10000_1.fastq.gz"), d("casava.1.8_2.fastq.gz")], [d("c18_10000_1.fastq_missing.gz"), d("c18_10000_2.fastq.gz")]]
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(DATA, "*.fastq.gz")))
    for n in names:
        if "_1." in n and n.replace("_1.", "_2.") in names and os.path.getsize(os.path.join(DATA, n)) < 200_000:
            m = n.replace("_1.", "_2.")
            for extra in ([], ["sorted"]):
                jobs.append([d(n), d(m)] + extra)
                jobs.append([d(m), d(n)] + extra)
    for a, b in (("syn_fp_1.fastq", "syn_fp_2.fastq"), ("syn_fp_2.fastq", "syn_fp_1.fastq"), ("syn_fp_1.fastq", "syn_fp_1.fastq"),
                 ("syn_fp_1.fastq", "empty.fastq"), ("empty.fastq", "syn_fp_1.fastq"), ("syn_fp_1.fastq", "syn_fp_2_trunc.fastq"),
                 ("syn_fp_1.fastq", "syn_fp_2_noat.fastq"), ("test_e9.fastq.gz", "test_1.fastq.gz"),
                 ("test_1.fastq.gz", "test_e3.fastq.gz"), ("test_e3.fastq.gz", "test_1.fastq.gz")):
        for extra in ([], ["sorted"]):
            jobs.append([d(a), d(b)] + extra)
    seen, out = set(), []
    for j in jobs:
        if tuple(j) not in seen:
            seen.add(tuple(j))
            out.append(j)
    return out + [[], ["a"], [d("test_1.fastq.gz"), d("test_1.fastq.gz"), "O1", "O2"]]


def gen_filterpair():
    import gzip
    import hashlib
    import tempfile

    if not os.path.exists(FP_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    s1, s2 = filterpair_synthetic()
    with open(os.path.join(DATA, "syn_fp_1.fastq"), "wb") as f:
        f.write(s1)
    with open(os.path.join(DATA, "syn_fp_2.fastq"), "wb") as f:
        f.write(s2)
    with open(os.path.join(DATA, "syn_fp_2_trunc.fastq"), "wb") as f:
        f.write(s2[: len(s2) // 2].rsplit(b"\n+\n", 1)[0] + b"\n")  # ends after a sequence line
    with open(os.path.join(DATA, "syn_fp_2_noat.fastq"), "wb") as f:
        recs = s2.split(b"\n@SYN")
        recs[120] = b"XSYN" + recs[120]  # a header without '@' in the middle of file 2
        f.write(recs[0] + b"".join((b"\n" + r) if r.startswith(b"XSYN") else (b"\n@SYN" + r) for r in recs[1:]))

    def pack(text):
        return {"sha256": hashlib.sha256(text).hexdigest(), "len": len(text),
                "text": text.decode("latin-1") if len(text) <= 40000 else None}
    out = []
    for args in filterpair_jobs():
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = list(args)
            if len(real) in (2, 3):
                real = real[:2] + [rel + "/p1.fastq.gz", rel + "/p2.fastq.gz", rel + "/up.fastq.gz"] + real[2:]
            real = [rel + "/" + a if a in ("O1", "O2") else a for a in real]
            p = subprocess.run(["fastq_filterpair"] + real, executable=FP_BIN, cwd=GOLD, capture_output=True, timeout=300)
            files = {}
            if p.returncode in (0, 3):
                for k in ("p1", "p2", "up"):
                    path = os.path.join(tmp, k + ".fastq.gz")
                    if os.path.exists(path) and p.returncode == 0:
                        files[k] = pack(gzip.decompress(open(path, "rb").read()))
            out.append({"args": args, "exit": p.returncode, "stdout": p.stdout.decode("latin-1"),
                        "stderr": p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), "files": files})
    with open(os.path.join(GOLD, "filterpair.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("filterpair invocations:", len(out), "by exit status:", by)


# ---- bam_add_tags (run_tests.sh:485-499) ---------------------------------------------------------------------------
BT_BIN = os.path.join(REPO, "oracle", "_ref", "bam_add_tags")
BT_DATA = os.path.join(GOLD, "data_tags")


def bam_tags_synthetic():
    """Seeded BAMs (tests/bamgen.py) whose read names stop get_barcodes (src/bam_add_tags.c:43-99) at every one of its
    exits: no STAGS_, a wrong CELL= / UMI= / SAMPLE= keyword, empty and non-empty values, names of other shapes mixed
    in, unmapped reads (tid -1: no tx tag), records that already carry aux tags, reads on several references."""
    import numpy as np

    sys.path.insert(0, REPO)
    from tests import bamgen

    rng = np.random.default_rng(4242)
    refs = tuple((b"ENST%011d" % (1000 + i), 5000 + i) for i in range(40))

    def bc(n):
        return bamgen.barcode(rng, n) if n else b""

    recs = []
    shapes = [
        lambda c, u, s, t: b"STAGS_CELL=%s_UMI=%s_SAMPLE=%s_ETAGS_%s" % (c, u, s, t),
        lambda c, u, s, t: b"STAGS_CELL=%s_UMI=%s_SAMPLE=%s_" % (c, u, s),          # nothing behind the last '_'
        lambda c, u, s, t: t,                                                        # a plain read name
        lambda c, u, s, t: b"STAGS_" + t + b"_x",                                    # STAGS_ but no CELL=
        lambda c, u, s, t: b"STAGS_CELL=%s_UMX=%s_SAMPLE=%s_ETAGS_%s" % (c, u, s, t),
        lambda c, u, s, t: b"STAGS_CELL=%s_UMI=%s_SAMPLX=%s_ETAGS_%s" % (c, u, s, t),
        lambda c, u, s, t: b"XSTAGS_CELL=%s_UMI=%s_SAMPLE=%s_ETAGS_%s" % (c, u, s, t),
        lambda c, u, s, t: b"STAGS_CELL=%s_UMI=%s_SAMPLE=%s_ETAGS_STAGS_CELL=AA_UMI=CC_SAMPLE=GG_ETAGS_%s" % (c, u, s, t),
    ]
    for i in range(3000):
        c, u, s = bc(int(rng.choice([0, 0, 8, 16, 30]))), bc(int(rng.choice([0, 6, 10, 12, 49]))), bc(int(rng.choice([0, 0, 0, 8])))
        tail = b"R%d:%d:%d" % (i, int(rng.integers(0, 99999)), int(rng.integers(0, 9)))
        shape = shapes[0] if rng.random() < 0.7 else shapes[int(rng.integers(0, len(shapes)))]
        name = shape(c, u, s, tail)[:250]
        aux = b""
        if rng.random() < 0.4:
            aux += bamgen.aux_int(b"NH", int(rng.integers(1, 4)))
        if rng.random() < 0.3:
            aux += bamgen.aux_z(b"XS", b"old_tag")
        tid = -1 if rng.random() < 0.1 else int(rng.integers(0, len(refs)))
        recs.append(bamgen.record(name, aux, tid=tid, flag=4 if tid < 0 else 0, seq_len=int(rng.integers(1, 120))))
    stream = bamgen.header(refs) + b"".join(recs)
    # gene <tab> transcript; a transcript listed twice (the first line wins), transcripts that are not references,
    # references without a line
    lines = []
    for i, (name, _) in enumerate(refs):
        if i % 5 != 4:
            lines.append(b"GENE%05d\t%s\n" % (i // 3, name))
    lines.insert(7, b"GENEDUP\t%s\n" % refs[2][0])
    lines.append(b"GENEX\tENSTnot_a_reference\n")
    return {"syn_tags.bam": bamgen.bgzf(stream, level=4), "syn_tags_map.tsv": b"".join(lines),
            "syn_tags_badmap.tsv": b"GENE1\tENST1\nonly_one_field\n"}


def bam_tags_jobs():
    d = lambda n: "data_tags/" + n
    jobs = []
    for bam, mp in (("trans_small.bam", "mapTrans2Gene.tsv"), ("syn_tags.bam", "syn_tags_map.tsv")):
        for extra in ([], ["--10x"], ["--tx"], ["--tx", "--tx_2_gx", d(mp)], ["--10x", "--tx", "--tx_2_gx", d(mp)]):
            jobs.append(["--inbam", d(bam), "--outbam", "OUT"] + extra)
        jobs.append(["--inbam", d(bam), "--outbam", "-", "--tx"])
    jobs.append(["--inbam", d("syn_tags.bam"), "--outbam", "OUT", "--tx", "--tx_2_gx", d("syn_tags_badmap.tsv")])
    # run_tests.sh:493-499
    jobs.append(["--inbam", d("trans_small.bam"), "--outbam", "OUT", "--tx", "--tx_2_gx", "aaaa" + d("mapTrans2Gene.tsv")])
    jobs.append([])
    jobs.append(["--inbam", d("trans_small.bam"), "--outbam", "OUT", "--tx_2_gx", d("mapTrans2Gene.tsv")])
    jobs.append(["--inbam", d("trans_small.bam_missing"), "--outbam", "OUT", "--tx", "--tx_2_gx", d("mapTrans2Gene.tsv")])
    jobs.append(["--inbam", d("trans_small.bam"), "--outbam", "folder/does/not/exist/tmp.bam", "--tx"])
    jobs.append(["--help"])
    jobs.append(["--inbam", d("trans_small.bam")])
    return jobs


def gen_bam_tags():
    import hashlib
    import tempfile

    if not os.path.exists(BT_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(BT_DATA, exist_ok=True)
    for n in ("trans_small.bam", "mapTrans2Gene.tsv"):   # the reference's own fixtures for this program
        shutil.copyfile(os.path.join(REF, "tests", n), os.path.join(BT_DATA, n))
        os.chmod(os.path.join(BT_DATA, n), 0o644)
    for n, b in bam_tags_synthetic().items():
        with open(os.path.join(BT_DATA, n), "wb") as f:
            f.write(b)
    out = []
    for args in bam_tags_jobs():
        with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
            rel = os.path.relpath(tmp, GOLD)
            real = [rel + "/o.bam" if a == "OUT" else a for a in args]
            p = subprocess.run(["bam_add_tags"] + real, executable=BT_BIN, cwd=GOLD, capture_output=True, timeout=300)
            entry = {"args": args, "exit": p.returncode, "stderr": p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/")}
            # the BAM written (file or stdout): its INFLATED bytes are what a reader sees
            path = os.path.join(tmp, "o.bam")
            blob = open(path, "rb").read() if os.path.exists(path) else (p.stdout if "-" in args else None)
            if blob is not None and p.returncode == 0:
                data = gzip.decompress(blob)
                entry["out_sha256"] = hashlib.sha256(data).hexdigest()
                entry["out_bytes"] = len(data)
            entry["out_created"] = os.path.exists(path)
            entry["stdout_is_bam"] = "-" in args
            if "-" not in args:
                entry["stdout"] = p.stdout.decode("latin-1")
            out.append(entry)
    with open(os.path.join(GOLD, "bam_tags.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("bam_add_tags invocations:", len(out), "by exit status:", by)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "fastq_info"):
        main()
    if which in ("all", "pre_barcodes"):
        gen_pre_barcodes()
    if which in ("all", "umi"):
        gen_umi()
    if which in ("all", "filterpair"):
        gen_filterpair()
    if which in ("all", "filters"):
        gen_filters()
    if which in ("all", "bam_tags"):
        gen_bam_tags()
