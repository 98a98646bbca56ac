"""TEST INFRASTRUCTURE, not product code: sequential restatement of the reference's bam_add_tags
(reference src/bam_add_tags.c, 0.25.3, on libbam 0.1.19: bam_read1 / bam_aux_append / bam_write1) on in-memory
BAM files - the step between fastq_pre_barcodes and bam_umi_count (sh/fastq2bam:116-273): the barcodes that
fastq_pre_barcodes put into the read names (STAGS_CELL=.._UMI=.._SAMPLE=.._ETAGS_) become aux tags.

Pinned: tests/test_oracle_bam_tags.py requires the exit status, stderr and the INFLATED output BAM stream of
every golden invocation in tests/golden/bam_tags.json (captured by tools/gen_golden.py from
oracle/_ref/bam_add_tags, the reference program compiled from its own sources; run_tests.sh:485-499).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import struct

from oracle.umi_oracle import bgzf_inflate

VERSION = "0.25.3"
MAX_BARCODE_LENGTH = 50   # src/fastq.h (the arrays get_barcodes writes into)
MAX_FEAT_LEN = 50         # src/bam_add_tags.c:36
USAGE = ("Usage: bam_add_tags --inbam <in.bam> --outbam <out.bam or - for stdout> [--tx] "
         "[--tx2gx map_file_gene_2_trans.tsv]")


class Undefined(Exception):
    """the reference reads or writes memory it does not own here (no defined behaviour to restate)"""


def get_barcodes(data: bytes, s: int, end: int):
    """src/bam_add_tags.c:43-99 on the C string at data[s:] (the qname; the scans for '_' do not stop at its NUL and
    run on through the bytes of the record).  (ok, cell, umi, sample).  `end`: end of the record's data."""
    def at(i):
        if i >= end:
            raise Undefined("get_barcodes reads past the record")
        return data[i]

    def expect(i, text):
        for k, c in enumerate(text):
            if at(i + k) != c:   # (|| chains: the first mismatch ends the evaluation)
                return False
        return True

    def value(i):
        z = i
        while at(z) != 0x5F:
            z += 1
        if z - i >= MAX_BARCODE_LENGTH:
            raise Undefined("barcode longer than the 50-byte arrays")
        return data[i:z], z + 1

    if not expect(s, b"STAGS_"):
        return False, b"", b"", b""
    i = s + 6
    if not expect(i, b"CELL="):
        return False, b"", b"", b""
    cell, i = value(i + 5)
    if not expect(i, b"UMI="):
        return False, cell, b"", b""
    umi, i = value(i + 4)
    if not expect(i, b"SAMPLE="):
        return False, cell, umi, b""
    sample, i = value(i + 7)
    return True, cell, umi, sample


def parse_header(stream: bytes):
    """bam_header_read (bam.c): magic, text, references.  (names, offset of the first alignment) or None"""
    if len(stream) < 12 or stream[:4] != b"BAM\x01":
        return None
    l_text = struct.unpack_from("<i", stream, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, p)[0]
    p += 4
    names = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", stream, p)[0]
        raw = stream[p + 4:p + 4 + l_name]
        names.append(raw.split(b"\0")[0])   # header->target_name[i] as a C string
        p += 4 + l_name + 4
    return names, p


def add_tags_stream(stream: bytes, tenx=False, tx_tag=False, tmap=None):
    """the alignment loop, src/bam_add_tags.c:250-294: the inflated input -> the inflated output, number of alignments"""
    names, p = parse_header(stream)
    out = [stream[:p]]                      # bam_header_write writes what bam_header_read read
    n = 0
    while p + 4 <= len(stream):
        block = struct.unpack_from("<i", stream, p)[0]
        if block < 32 or p + 4 + block > len(stream):
            break                           # bam_read1 < 0
        rec = stream[p + 4:p + 4 + block]
        tid = struct.unpack_from("<i", rec, 0)[0]
        add = b""
        ok, cell, umi, sample = get_barcodes(stream, p + 4 + 32, p + 4 + block)
        if ok:
            def z(tag, val):
                return tag + b"Z" + val + b"\0"    # bam_aux_append(.., 'Z', len + 1, ..)
            if umi:
                add += z(b"UB" if tenx else b"RX", umi)   # GET_UMI_TAG, src/sam_tags.h:40-47
            if cell:
                add += z(b"CR", cell)
            if sample:
                add += z(b"BC", sample)
            if tx_tag and tid >= 0:
                if tid >= len(names):
                    raise Undefined("tid beyond the header's references")
                tx = names[tid]
                add += z(b"tx", tx)
                if tmap is not None and tx in tmap:
                    add += z(b"GX", tmap[tx])
        out.append(struct.pack("<i", block + len(add)) + rec + add)
        p += 4 + block
        n += 1
    return b"".join(out), n


def load_map(text: bytes):
    """src/bam_add_tags.c:203-232: gene <tab> transcript lines (fgets of 999 bytes, strtok on tab/newline); the first
    line of a transcript wins (hash.c:161-184 appends, get_gene returns the first match).  (map, entries) or an error"""
    tmap, n, p = {}, 0, 0
    while p < len(text):
        e = text.find(b"\n", p, p + 999)
        chunk = text[p:e + 1] if e >= 0 else text[p:p + 999]
        p += len(chunk)
        line = chunk.split(b"\0")[0]
        if not line:
            continue
        toks = [t for t in line.replace(b"\n", b"\t").split(b"\t") if t]
        if len(toks) < 2:
            # what the message prints: the buffer after strtok put a NUL behind the first token (if there is one)
            k = 0
            while k < len(line) and line[k] in b"\t\n":
                k += 1
            e = k
            while e < len(line) and line[e] not in b"\t\n":
                e += 1
            return None, (line[:e] if e > k else line)
        gx, tx = toks[0], toks[1]
        if len(gx) >= MAX_FEAT_LEN or len(tx) >= MAX_FEAT_LEN:
            raise Undefined("strcpy into the 50-byte fields of TGM")
        tmap.setdefault(tx, gx)
        n += 1
    return tmap, n


def run_bam_add_tags(argv, reader, writable=lambda path: True):
    """main(): argv without the program name; reader(path) -> file bytes or None; writable(path): can the output be
    created.  Returns exit, stderr (str), stdout (bytes: the inflated BAM when --outbam -), files {path: inflated BAM,
    or None for an output that was created and never written}."""
    err, files = [], {}
    inbam = outbam = mapf = None
    tenx = tx = hlp = False
    i = 0
    while i < len(argv):           # getopt_long over the table at :146-155 (exact long names, separate values)
        a = argv[i]
        if a in ("--inbam", "-i"):
            inbam = argv[i + 1]; i += 1
        elif a in ("--outbam", "-o"):
            outbam = argv[i + 1]; i += 1
        elif a in ("--tx_2_gx", "-m"):
            mapf = argv[i + 1]; i += 1
        elif a == "--tx":
            tx = True
        elif a in ("--10x", "-X"):
            tenx = True
        elif a in ("--help", "-h"):
            hlp = True
        elif a == "--verbose":
            pass
        i += 1

    def done(status, stdout=b""):
        return {"exit": status, "stderr": "".join(err), "stdout": stdout, "files": files}

    def print_error(msg):          # PRINT_ERROR, src/fastq.h
        err.append("\nERROR: " + msg + "\n")

    if hlp:
        err.append(USAGE + "\n")
        return done(0)
    if inbam is None or outbam is None:
        print_error(USAGE)
        return done(1)
    if not tx and mapf is not None:
        print_error("missing  --tx when --tx_2_gx is provided\n")
        print_error(USAGE)
        return done(1)   # PARAMS_ERROR_EXIT_STATUS
    raw = reader(inbam)
    # (the output is opened - created - before the input is looked at, :189-199)
    out_ok = outbam == "-" or writable(outbam)
    if out_ok and outbam != "-":
        files[outbam] = None
    if raw is None:
        err.append("open: No such file or directory\n")   # bgzf.c: perror("open")
        print_error("Failed to open BAM file %s" % inbam)
        return done(1)
    if not out_ok:
        print_error("Failed to open BAM file %s" % outbam)
        return done(1)
    tmap = None
    if mapf is not None:
        text = reader(mapf)
        if text is None:
            print_error("Failed to open file %s" % mapf)
            return done(1)
        tmap, n = load_map(text)
        if tmap is None:
            print_error("Failed to find the gene and transcript ids in %s\n" % n.decode("latin-1"))
            return done(1)
        err.append("unique gene/transcript pairs %d\n" % n)
    to_stdout = outbam == "-"
    stream = bgzf_inflate(raw)
    if not to_stdout:
        err.append("bam_add_tags version %s\n" % VERSION)
        err.append("Processing %s\n" % inbam)
    out, _ = add_tags_stream(stream, tenx, tx, tmap)
    if not to_stdout:
        files[outbam] = out
        err.append("Processing %s complete\n" % inbam)
    return done(0, out if to_stdout else b"")
