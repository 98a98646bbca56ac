"""TEST INFRASTRUCTURE, not product code: sequential restatement of the reference's
fastq_pre_barcodes (reference src/fastq_pre_barcodes.c, 0.25.3) on in-memory file images.

Pinned: tests/test_oracle_pre_barcodes.py requires identical exit status, stdout, stderr and
decompressed output files for every golden invocation in tests/golden/pre_barcodes.json (captured
from oracle/_ref/fastq_pre_barcodes, the reference program compiled from its own sources).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pure-Python loops: meant for small inputs.  Each function names the reference lines it follows.
"""
import re

MAX_LABEL, MAX_READ = 1000, 2500000
READ1, READ2, INDEX1, INDEX2, INDEX3 = 1, 2, 3, 4, 5
UNDEF = -1
REFS = {"read1": 1, "read2": 2, "index1": 3, "index2": 4, "index3": 5}


class Exit(Exception):
    def __init__(self, status):
        self.status = status


class Run:
    def __init__(self):
        self.out, self.err = [], []

    def eprint(self, s):
        self.err.append(s)

    def error(self, s):  # PRINT_ERROR, src/fastq.h:69
        self.err.append("\nERROR: " + s + "\n")

    def info(self, s):  # PRINT_INFO, src/fastq.h:68
        self.err.append("INFO:" + s + "\n")


def cstr(b):
    """bytes up to the first NUL: what a C string function sees"""
    i = b.find(b"\0")
    return b if i < 0 else b[:i]


class MemFile:
    """FASTQ_FILE over an image; gzgets/gzeof semantics (src/fastq.c:202-261)"""

    def __init__(self, name, image):
        self.name, self.img, self.pos, self.past = name, image, 0, False
        self.cline = 0
        self.fmt = None  # readname_format
        self.space = None

    def gets(self, limit):
        if self.pos >= len(self.img):
            self.past = True
            return b""
        end = self.img.find(b"\n", self.pos, self.pos + limit - 1)
        if end < 0:
            stop = min(len(self.img), self.pos + limit - 1)
            if stop == len(self.img) and stop - self.pos < limit - 1:
                self.past = True
        else:
            stop = end + 1
        s = self.img[self.pos:stop]
        self.pos = stop
        return s


class Entry:
    def __init__(self):
        self.hdr1 = self.hdr2 = self.seq = self.qual = b""
        self.read_len = 0


def read_next_entry(run, f, e):
    """fastq_read_next_entry -> fastq_read_entry, src/fastq.c:237-261"""
    if f.past:
        return 0
    e.hdr1 = f.gets(MAX_LABEL)
    if cstr(e.hdr1) == b"":
        return 0
    e.seq = f.gets(MAX_READ)
    e.hdr2 = f.gets(MAX_LABEL)
    e.qual = f.gets(MAX_READ)
    if cstr(e.seq) == b"" or cstr(e.hdr2) == b"" or cstr(e.qual) == b"":
        run.error("Error in file %s: line %d: file truncated" % (f.name, f.cline))
        raise Exit(1)
    f.cline += 4
    e.read_len = len(cstr(e.seq))
    return 1


def get_readname(run, f, e):
    """fastq_get_readname(fd, e, rn, &len, TRUE) with is_pe set, src/fastq.c:442-516"""
    hdr = e.hdr1
    if hdr[:1] != b"@":
        run.error("Error in file %s: line %d: wrong header %s" % (f.name, f.cline, cstr(hdr).decode("latin-1")))
        raise Exit(3)
    rn = cstr(hdr[1:])[: MAX_LABEL - 1]
    if f.fmt is None:
        if re.search(rb"[A-Z0-9:]* [1234]:[YN]:[0-9]*.*", rn):
            run.eprint("CASAVA=1.8\n")
            f.fmt = 1
        elif re.search(rb"^[0-9]+[\n\r]?\Z", rn):
            run.eprint("Read name provided as an integer\n")
            f.fmt = 2
        elif not re.search(rb"[# \t/:][0-9abAB][\n\r]?\Z", rn):
            run.eprint("Read name provided with no suffix\n")
            f.fmt = 2
        else:
            f.fmt = 0
    if f.space is None:
        f.space = 1 if re.search(rb"^[GT]?[0123n.NtT]+\n?\Z", cstr(e.seq)) else 0
        if f.space == 1:
            run.eprint("Color space\n")
    if f.fmt == 0:
        n = len(rn) - 1  # is_pe
        return rn[: n - 1] if n >= 1 else rn
    if f.fmt == 2:
        n = len(rn)
        return rn[: n - 1] if n >= 1 else rn
    sp = rn.find(b" ")
    if sp < 0:
        sp = len(rn)
    rn = rn[:sp]
    if sp >= 2 and rn[sp - 2:sp - 1] == b"/":
        rn = rn[: sp - 2]
    return rn


def get_barcode(run, e, P, read, offset, size):
    """src/fastq_pre_barcodes.c:218-259; returns (ok, seq, qual)"""
    if read == UNDEF or offset == UNDEF or size == 0:
        return True, b"", b""
    rl1 = e.read_len - 1
    off_u = offset if offset >= 0 else offset + (1 << 64)  # long compared with unsigned long
    end_u = (offset + size) if (offset + size) >= 0 else offset + size + (1 << 64)
    if rl1 < 0:
        rl1 += 1 << 64
    if off_u > rl1 or end_u > rl1:
        run.eprint("Warning: Read too short - barcode not found\n")
        return False, b"", b""
    if P["min_qual"] > 0:
        q = e.qual
        for x in range(offset, offset + size):
            c = q[x] if x < len(q) else 0
            if c >= 128:
                c -= 256
            if c - P["phred_encoding"] < P["min_qual"]:
                return False, b"", b""
    return True, cstr(e.seq[offset:offset + size]), cstr(e.qual[offset:offset + size])


def add_tags2readname(e, cell, umi, sample):
    """src/fastq_pre_barcodes.c:192-216"""
    if not (cell or umi or sample):
        return
    tags = b"STAGS_CELL=" + cell + b"_UMI=" + umi + b"_SAMPLE=" + sample + b"_ETAGS_"
    e.hdr1 = e.hdr1[:1] + tags + cstr(e.hdr1[1:])
    e.hdr2 = e.hdr2[:1] + b"\n"


def shift_cut(s, offset, size, read_len):
    """what slice_read leaves in seq / qual (src/fastq_pre_barcodes.c:168-189), as a C string"""
    s = cstr(s)
    if size == 0:
        return b"\n"
    if offset > 0:
        n = size if size != -1 else read_len
        # seq[x] = seq[x+offset] for x in 0..n: bytes past the old terminator are whatever the
        # buffer held; only reachable for nonsensical parameters, modelled as NULs
        src = s + b"\0" * (offset + n + 2)
        buf = bytearray(src)
        for x in range(0, n + 1):
            buf[x] = src[x + offset]
        s = bytes(buf)
    if size == -1:
        return b""  # seq[-1]='\n', seq[0]='\0'
    buf = bytearray(s + b"\0" * (size + 2))
    buf[size] = 0x0A
    buf[size + 1] = 0
    return cstr(bytes(buf))


def slice_read(e, P, x):
    """src/fastq_pre_barcodes.c:160-190"""
    off, size = P["read_offset"][x], P["read_size"][x]
    if off == UNDEF:
        return
    if x < INDEX1 and off == 0 and size == -1:
        return
    e.hdr2 = e.hdr2[:1] + b"\n"
    e.seq = shift_cut(e.seq, off, size, e.read_len)
    e.qual = shift_cut(e.qual, off, size, e.read_len)


def parse_args(run, argv):
    """getopt_long loop, src/fastq_pre_barcodes.c:360-528 (exact long names only)"""
    P = {"file": [None] * 6, "outfile": [None] * 3, "phred_encoding": 64, "read_offset": [UNDEF] * 3,
         "read_size": [0] * 3, "cell_read": UNDEF, "cell_offset": UNDEF, "cell_size": 0, "sample_read": UNDEF,
         "sample_offset": UNDEF, "sample_size": 0, "umi_read": UNDEF, "umi_offset": UNDEF, "umi_size": 0,
         "interleaved": [0, 0, 0], "has_interleaved": False, "min_qual": 0, "num_input_files": 0,
         "sam": False, "help": False, "tenx": False}

    def ref(s):
        if s not in REFS:
            run.error("invalid file reference %s (valid values are read1,read2, index1,index2,index3)\n" % s)
            raise Exit(1)
        return REFS[s]

    def set_file(name, idx):
        if P["file"][idx] is None:
            P["num_input_files"] += 1
        P["file"][idx] = name

    flags = {"--sam": ("sam", True), "--fastq": ("sam", False), "--help": ("help", True), "--10x": ("tenx", True),
             "-X": ("tenx", True)}
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in flags:
            P[flags[a][0]] = flags[a][1]
            i += 1
            continue
        if a in ("--verbose", "--brief", "--paired_end", "--single_end"):
            i += 1
            continue
        if not a.startswith("-") or i + 1 >= len(argv):
            i += 1
            continue
        v = argv[i + 1]
        i += 2
        if a == "--interleaved":
            toks = [t for t in v.split(",") if t != ""]
            xx = 0
            for t in toks:
                P["interleaved"][xx] = ref(t)
                xx += 1
                if xx == 3:
                    break
            if xx != 2:
                run.error("two file references should be passed to --interleaved")
                raise Exit(1)
            P["has_interleaved"] = True
        elif a == "--umi_read":
            P["umi_read"] = ref(v)
        elif a == "--cell_read":
            P["cell_read"] = ref(v)
        elif a == "--sample_read":
            P["sample_read"] = ref(v)
        elif a in ("--umi_offset", "--umi_size", "--cell_offset", "--cell_size", "--sample_offset", "--sample_size",
                   "--min_qual", "--phred_encoding"):
            P[a[2:]] = int(v)
        elif a in ("--read1_offset", "--read2_offset"):
            P["read_offset"][int(a[6])] = int(v)
        elif a in ("--read1_size", "--read2_size"):
            P["read_size"][int(a[6])] = int(v)
        elif a in ("--read1", "--read2", "--index1", "--index2", "--index3"):
            set_file(v, REFS[a[2:]])
        elif a in ("--outfile1", "--outfile2"):
            P["outfile"][int(a[9])] = v
    return P


USAGE_MSG = None  # the help text is product documentation; the oracle only checks the status


def run_pre_barcodes(argv, read_image, argv0="fastq_pre_barcodes"):
    """Returns dict(exit, stdout, stderr, files={1: bytes, 2: bytes}).  read_image(name) -> bytes."""
    run = Run()
    files_out = {}
    status = 0
    try:
        run.eprint("fastq_utils 0.25.3\n")
        P = parse_args(run, argv)
        if P["help"]:
            run.eprint("usage: fastq_pre_barcodes --read1 fastq_file --outfile1 out_file [optional parameters]\n")
            run.eprint("<help text>\n")
            raise Exit(0)
        run.info("Validating options...")
        if P["file"][1] is None:
            run.error("missing input file (-read1)")
            raise Exit(1)
        if P["outfile"][1] is None:
            run.error("if single_end then -outfile1 should be provided")
            raise Exit(1)
        run.info("Options OK.")
        run.info("input files %d" % P["num_input_files"])
        fd = [None] * 6
        m = [Entry() for _ in range(6)]
        for x in range(1, 6):
            if P["file"][x] is not None:
                fd[x] = MemFile(P["file"][x], read_image(P["file"][x]))
        if not P["sam"]:
            for x in (1, 2):
                if P["outfile"][x] is not None:
                    files_out[x] = []
        else:
            run.out.append("@HD\tVN:1.0 SO:unknown\n")
            run.out.append("@PG\tID:1 PN:fastq_pre_barcodes CL:%s" % argv0)
            for a in argv[:-1]:
                run.out.append(" %s" % a)
            run.out.append("\n")
        umi_tag, umi_qtag = ("UB", "UY") if P["tenx"] else ("RX", "QX")
        processed = discarded = 0

        def any_eof():
            return any(fd[x] is not None and fd[x].past for x in range(1, 6))

        done = False
        while not any_eof() and not done:
            for x in range(1, 6):
                if fd[x] is not None and read_next_entry(run, fd[x], m[x]) == 0:
                    done = True
                    break
            if done:
                break
            if P["has_interleaved"]:
                k = P["interleaved"][1]
                if read_next_entry(run, fd[k], m[k]) == 0:
                    break
            if P["num_input_files"] > 1:
                names = {}
                for x in range(1, 6):
                    if fd[x] is not None:
                        names[x] = get_readname(run, fd[x], m[x])
                for x in (2, 3, 4, 5):
                    if fd[x] is not None and names[1] != names[x]:
                        run.error("Readnames do not match across files (read #%d)" % (processed + 1))
                        raise Exit(3)
            processed += 1
            cell = umi = sample = b""
            cellq = umiq = sampleq = b""
            skip = False
            for x in range(1, 6):
                if fd[x] is None:
                    continue
                ok = True
                if P["umi_read"] == x:
                    ok, s, q = get_barcode(run, m[x], P, x, P["umi_offset"], P["umi_size"])
                    if ok:
                        umi, umiq = s, q
                if ok and P["sample_read"] == x:
                    ok, s, q = get_barcode(run, m[x], P, x, P["sample_offset"], P["sample_size"])
                    if ok:
                        sample, sampleq = s, q
                if ok and P["cell_read"] == x:
                    ok, s, q = get_barcode(run, m[x], P, x, P["cell_offset"], P["cell_size"])
                    if ok:
                        cell, cellq = s, q
                if not ok:
                    discarded += 1
                    skip = True
                    break
            if not skip:
                if P["sam"]:
                    se = fd[2] is None
                    for mate, x in ((1, 1), (2, 2)):
                        if mate == 2 and se:
                            break
                        flag = 4 if se else (4 | 8 | 1 | (0x40 if mate == 1 else 0x80))
                        slice_read(m[x], P, x)
                        seq, qual = cstr(m[x].seq), cstr(m[x].qual)
                        ln = len(seq)
                        seq, qual = seq[: ln - 1] if ln else seq, qual[: len(qual) - 1] if qual else qual
                        h = bytearray(cstr(m[x].hdr1))
                        for i, c in enumerate(h):  # format_read_name, src/fastq_pre_barcodes.c:300-308
                            if c == 0x20:
                                h[i] = 0x40
                        rn = cstr(bytes(h).replace(b"\n", b"\0"))[1:]
                        shown = (ln - 1) if mate == 1 else ln
                        o = "%d\t%d\t*\t0\t255\t*\t*\t0\t%d" % (processed, flag, shown & 0xFFFFFFFF)
                        o += "\t%s\t%s\ton:Z:%s" % (seq.decode("latin-1"), qual.decode("latin-1"), rn.decode("latin-1"))
                        o += "\top:Z:%s" % qual.decode("latin-1")
                        if umi:
                            o += "\t%s:Z:%s\t%s:Z:%s" % (umi_tag, umi.decode("latin-1"), umi_qtag, umiq.decode("latin-1"))
                        if cell:
                            sep = "\t" if mate == 1 else " "  # src/fastq_pre_barcodes.c:705 prints a blank
                            o += "%sCR:Z:%s\tCY:Z:%s" % (sep, cell.decode("latin-1"), cellq.decode("latin-1"))
                        if sample:
                            o += "\tBC:Z:%s\tQT:Z:%s" % (sample.decode("latin-1"), sampleq.decode("latin-1"))
                        run.out.append(o + "\n")
                else:
                    for x in (1, 2):
                        if x in files_out:
                            add_tags2readname(m[x], cell, umi, sample)
                            slice_read(m[x], P, x)
                            files_out[x].append(cstr(m[x].hdr1) + cstr(m[x].seq) + cstr(m[x].hdr2) + cstr(m[x].qual))
                c = fd[1].cline // 4
                if c % 100000 == 0:
                    run.eprint("\b" * 15 + "%d" % c)
            if P["has_interleaved"] and not skip:
                # (`continue` on a discarded read skips this re-synchronising read too:
                # src/fastq_pre_barcodes.c:653 vs :722-725)
                k = P["interleaved"][0]
                if read_next_entry(run, fd[k], m[k]) == 0:
                    break
        run.info("Reads processed: %d" % processed)
        run.info("Reads discarded: %d" % discarded)
    except Exit as e:
        status = e.status
    return {"exit": status, "stdout": "".join(run.out), "stderr": "".join(run.err),
            "files": {k: b"".join(v) for k, v in files_out.items()}}
