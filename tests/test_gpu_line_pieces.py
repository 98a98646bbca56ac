"""Lines the reference does not take whole - and what it does with them, reproduced by the four programs that COPY
records (fastq_filter_n, fastq_trim_poly_at, fastq_filterpair, fastq_pre_barcodes):

  * a line beyond the gzgets buffers (src/fastq.c:249-253: 999 bytes of a header line, 2 499 999 of a sequence / quality
    line) comes back in pieces, every piece the next field of the record, and the pieces are copied as they are;
  * a NUL byte inside a line ends the string the reference holds (strlen, gzputs, printf("%s")): the rest of the line and
    its '\\n' are gone; at the start of a sequence / second header / quality line it is an empty string - "file truncated".

Both are one thing to the kernels: a line is a C string (csrc/fqg_barcode_kernels.hip: bc_clip_nul), and a plain file with a
line beyond the limits is read once more by a child process that cuts it where gzgets cuts it (host/fq_respawn.h,
fq_reframe.h).  Everything is compared with the reference binaries (oracle/_ref), byte for byte: exit status, stdout,
stderr (without the progress ticker), decompressed output files."""
import gzip
import os
import subprocess
import tempfile

import pytest

from tests.test_oracle_vs_ref_fuzz import overlong_images
from tests.util import REPO, SideBySide, strip_progress, thinned

pytestmark = pytest.mark.gpu
REF = os.path.join(REPO, "oracle", "_ref")
BIN = os.path.join(REPO, "bin")
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "fastq_filter_n")), reason="oracle/_ref not built")


def images():
    ok = b"@r1 1:N:0:A\nACGTNAAAAA\n+\nIIIIIIIIII\n@r2 1:N:0:A\nTTTTTGCANN\n+\nIIIIIIIIII\n"
    out = dict(overlong_images())
    out["nul_in_seq"] = ok + b"@a 1:N:0:A\nAC\0GTAAAA\n+\nIIIIIIIII\n" + ok
    out["nul_in_qual"] = ok + b"@a 1:N:0:A\nACNNGTAAAA\n+\nII\0IIIIIII\n" + ok
    out["nul_in_hdr1"] = ok + b"@a\0zz 1:N:0:A\nACGT\n+\nIIII\n" + ok
    out["nul_in_hdr2"] = ok + b"@a 1:N:0:A\nACGT\n+a\0b\nIIII\n" + ok
    out["nul_ends_seq_before_newline"] = ok + b"@a 1:N:0:A\nACGTAAAAA\0\n+\nIIIIIIIII\n" + ok
    out["nul_starts_seq"] = ok + b"@a 1:N:0:A\n\0ACGT\n+\nIIII\n" + ok          # an empty string: file truncated
    out["nul_starts_hdr2"] = ok + b"@a 1:N:0:A\nACGT\n\0+\nIIII\n" + ok
    out["nul_starts_qual"] = ok + b"@a 1:N:0:A\nACGT\n+\n\0III\n" + ok
    out["nul_starts_record"] = ok + b"\0@a 1:N:0:A\nACGT\n+\nIIII\n" + ok         # "no entry": the loop ends, cleanly
    out["nul_in_the_last_unterminated_line"] = ok + b"@a 1:N:0:A\nACGT\n+\nII\0I"
    return out


IMAGES = images()
PROGRAMS = {
    "filter_n": ("fastq_filter_n", ["-n", "10", "IN"]),
    "trim_poly_at": ("fastq_trim_poly_at", ["--file", "IN", "--outfile", "o.fastq.gz", "--min_poly_at_len", "3", "--min_len", "2"]),
    "trim_poly_at_stdout": ("fastq_trim_poly_at", ["--file", "IN", "--outfile", "-", "--min_poly_at_len", "4"]),
    "filterpair": ("fastq_filterpair", ["IN", "IN", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"]),
    "filterpair_sorted": ("fastq_filterpair", ["IN", "IN", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz", "sorted"]),
    "pre_barcodes_fastq": ("fastq_pre_barcodes", ["--read1", "IN", "--index1", "IN", "--umi_read", "index1", "--umi_offset", "0",
                                                  "--umi_size", "4", "--phred_encoding", "33", "--min_qual", "1", "--outfile1", "o.fastq.gz"]),
    "pre_barcodes_sam": ("fastq_pre_barcodes", ["--index1", "IN", "--min_qual", "1", "--phred_encoding", "33", "--umi_read", "index1",
                                                "--umi_offset", "0", "--umi_size", "4", "--cell_read", "index1", "--cell_offset", "0",
                                                "--cell_size", "3", "--read1_offset", "0", "--read1_size", "-1", "--read1", "IN",
                                                "--outfile1", "-", "--sam"]),
}
HOW = {"plain_file": None, "gz_file": None, "small_pieces": {"FQGPU_CHUNK_MB": "1"},
       "several_devices": {"FQGPU_DEVICES": "0,0", "FQGPU_CHUNK_MB": "1", "FQGPU_BLOCK_RECORDS": "3"}}


def one_run(root, exe_dir, prog, which, how):
    """the program in a directory of its own: exit status, stdout (inflated when it is a gzip stream), stderr, output files"""
    argv0, args = PROGRAMS[prog]
    name = "f.fastq.gz" if how == "gz_file" else "f.fastq"
    d = tempfile.mkdtemp(dir=root)
    with open(os.path.join(d, name), "wb") as f:
        f.write(gzip.compress(IMAGES[which], 1) if how == "gz_file" else IMAGES[which])
    env = dict(os.environ, **(HOW[how] or {})) if exe_dir == BIN else dict(os.environ)
    p = subprocess.run([argv0] + [name if a == "IN" else a for a in args], executable=os.path.join(exe_dir, argv0), cwd=d,
                       capture_output=True, timeout=600, env=env)
    out = p.stdout
    if out[:2] == b"\x1f\x8b":
        out = gzip.decompress(out)
    files = {}
    for fn in sorted(os.listdir(d)):
        if fn != name:
            raw = open(os.path.join(d, fn), "rb").read()
            files[fn] = gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw
    return p.returncode, out, strip_progress(p.stderr.decode("latin-1")), files


ROOT = tempfile.TemporaryDirectory()
KEYS = [(prog, which, how) for prog in PROGRAMS for which in sorted(IMAGES) for how in HOW
        if not (how == "several_devices" and not prog.startswith("pre_barcodes"))   # (FQGPU_DEVICES: fastq_pre_barcodes only)
        and not (how == "small_pieces" and which.startswith("nul_"))]               # (the NUL images are a few hundred bytes)
# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
# (on a box that starts programs slowly the ways other than the plain file are thinned out: tests/util.py)
OURS = SideBySide(lambda k: one_run(ROOT.name, BIN, *k), KEYS,
                  select=lambda ks: [k for k in ks if k[2] == "plain_file"] + thinned([k for k in ks if k[2] != "plain_file"], key="-".join))


def ref_kind(key):  # what the reference is given: the file, plain or gzipped (the other ways are this program's business)
    return key[0], key[1], "gz_file" if key[2] == "gz_file" else "plain_file"


REF_KEYS = sorted({ref_kind(k) for k in KEYS})
THEIRS = SideBySide(lambda k: one_run(ROOT.name, REF, *k), REF_KEYS, workers=8)


@needs_ref
@pytest.mark.parametrize("key", KEYS, ids=["-".join(k) for k in KEYS])
def test_copy_programs_on_lines_the_reference_reads_in_pieces(key):
    got, want = OURS.get(key), THEIRS.get(ref_kind(key))
    assert got[0] == want[0], (got[2][-600:], want[2][-600:])
    assert got[2] == want[2]
    assert got[1] == want[1], (len(got[1]), len(want[1]))
    assert sorted(got[3]) == sorted(want[3])
    for fn in want[3]:
        assert got[3][fn] == want[3][fn], (fn, len(got[3][fn]), len(want[3][fn]))
