"""TEST INFRASTRUCTURE - CPU restatement of the reference's fastq_filter_n and fastq_trim_poly_at.

Not part of the product: only tests/ may import this.  Pinned on tests/golden/filters.json
(invocations of the reference binaries, tools/gen_golden.py filters), which includes the reference
suite's own golden poly_at_len3.fastq.gz (run_tests.sh:199).

  filter_n        reference src/fastq_filter_n.c:33-95
  trim_poly_at    reference src/fastq_trim_poly_at.c:77-119 (the trim) and :123-233 (the program)
Records are read as fastq_read_entry does (reference src/fastq.c:245-261): four gzgets lines kept
with their '\\n'; read_len = strlen(seq).
"""
import gzip

VERSION = b"fastq_utils 0.25.3\n"


MAX_LABEL_LENGTH = 1000      # src/fastq.h:36
MAX_READ_LENGTH = 2500000    # src/fastq.h:32


def _gzgets(data, pos, limit):
    """gzgets(fd, buf, limit): at most limit - 1 bytes, through the first '\n' -> (piece, new pos); b"" at the end of
    the file (GZ_READ then stores a NUL in buf[0], src/fastq.c:202-206)"""
    end = data.find(b"\n", pos, pos + limit - 1)
    stop = end + 1 if end >= 0 else min(len(data), pos + limit - 1)
    return data[pos:stop], stop


def _lines4(data):
    """fastq_read_entry until it returns 0 or exits (src/fastq.c:245-261) -> (records as 4-tuples of the pieces the
    four reads return, tail): tail != 0 = the reference ends with "file truncated" after these records.  A line longer
    than its read's buffer comes back in pieces, the next piece to the NEXT read; a record whose first read yields a
    string that starts with NUL (or nothing) ends the file quietly, one whose other reads do is the truncation."""
    records, pos = [], 0
    while pos < len(data):
        h1, pos = _gzgets(data, pos, MAX_LABEL_LENGTH)
        if h1[:1] == b"\0":
            break
        seq, pos = _gzgets(data, pos, MAX_READ_LENGTH)
        h2, pos = _gzgets(data, pos, MAX_LABEL_LENGTH)
        qual, pos = _gzgets(data, pos, MAX_READ_LENGTH)
        if any(x[:1] in (b"", b"\0") for x in (seq, h2, qual)):
            return records, 1
        records.append((h1, seq, h2, qual))
    return records, 0


def _cstr(b):
    i = b.find(b"\0")
    return b if i < 0 else b[:i]


def read_input(path, opener=None):
    raw = opener(path) if opener else open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return raw


def err(msg):
    return b"\nERROR: " + msg + b"\n"


def filter_n(argv, opener=None):
    """argv without the program name -> dict(exit, stdout, stderr); progress marks are not restated"""
    stderr = VERSION
    # getopt(argc, argv, "n:") with opterr = 0 (src/fastq_filter_n.c:45-58); GNU getopt permutes
    nopt, max_n, rest = 0, 0, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--":
            rest += argv[i + 1:]
            break
        if a.startswith("-") and len(a) > 1:
            j = 1
            while j < len(a):
                ch = a[j]
                if ch == "n":
                    val = a[j + 1:] if j + 1 < len(a) else (argv[i + 1] if i + 1 < len(argv) else None)
                    if val is None:
                        nopt += 1
                        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Option -n invalid")}
                    if j + 1 >= len(a):
                        i += 1
                    max_n = _atoi_unsigned(val)
                    if max_n > 100:
                        max_n = 100
                    nopt += 2
                    break
                return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Option -" + ch.encode("latin-1") + b" invalid")}
            i += 1
            continue
        rest.append(a)
        i += 1
    argc = len(argv) + 1
    if argc - nopt < 2 or argc - nopt > 3:
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Usage: fastq_filter_n [ -n 0 ] fastq1")}
    if max_n > 0:
        stderr += b"Discard reads with more than %d%% of Ns\n" % max_n
    else:
        stderr += b"Discard reads with at least one N\n"
    # argv[nopt+1] after getopt's permutation: options first, then the operands in order
    path = rest[0]
    try:
        data = read_input(path, opener)
    except (FileNotFoundError, OSError):
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Unable to open " + path.encode())}
    records, tail = _lines4(data)
    out = []
    for h1, seq, h2, qual in records:
        h1, seq, h2, qual = _cstr(h1), _cstr(seq), _cstr(h2), _cstr(qual)
        read_len = len(seq)
        max_num_n = (read_len * max_n // 100) & 0xFFFFFFFF
        body = seq.split(b"\n")[0]
        num_n = body.count(b"N") + body.count(b"n")
        if num_n <= max_num_n:
            out += [h1, seq, h2, qual]
    if tail:
        stderr += err(b"Error in file %s: line %d: file truncated" % (path.encode(), 4 * len(records)))
        return {"exit": 1, "stdout": b"".join(out), "stderr": stderr}
    return {"exit": 0, "stdout": b"".join(out), "stderr": stderr}


def _atoi_unsigned(s):
    s = s.strip()
    sign, k = 1, 0
    if s[:1] in ("+", "-"):
        sign = -1 if s[0] == "-" else 1
        k = 1
    d = ""
    while k < len(s) and s[k].isdigit():
        d += s[k]
        k += 1
    return (sign * int(d or "0")) & 0xFFFFFFFF


def _atol(s):
    s = s.strip()
    sign, k = 1, 0
    if s[:1] in ("+", "-"):
        sign = -1 if s[0] == "-" else 1
        k = 1
    d = ""
    while k < len(s) and s[k].isdigit():
        d += s[k]
        k += 1
    return sign * int(d or "0")


def trim_record(seq, qual, min_poly_at_len):
    """trim_poly_at (src/fastq_trim_poly_at.c:77-119) on C strings -> (seq', qual', read_len', trimmed)"""
    L, Lq = len(seq), len(qual)
    if min_poly_at_len <= 0:
        return seq, qual, L, False
    x = L - 2  # get_elength = read_len - 2 as a signed offset
    matched1 = 0
    while x >= 0 and seq[x:x + 1] in (b"N", b"A", b"n", b"a"):
        matched1 += 1
        x -= 1
    if matched1 >= min_poly_at_len:
        keep = x + 1
        nseq = seq[:keep] + b"\n"
        if keep <= Lq:
            nqual = qual[:keep] + b"\n"
        else:
            nqual = qual  # the stores land behind the string's terminator
        return nseq, nqual, L - matched1, True
    matched2 = 0
    for i in range(L):
        if seq[i:i + 1] not in (b"N", b"T", b"n", b"t"):
            break
        matched2 += 1
    if matched2 >= min_poly_at_len:
        nseq = seq[matched2:]
        if Lq < matched2:
            nqual = None  # stale buffer content in the reference: undefined
        elif Lq <= L:
            nqual = qual[matched2:]
        else:
            nqual = qual[matched2:L + 1] + qual[L - matched2 + 1:]
        return nseq, nqual, L - matched2, True
    return seq, qual, L, False


USAGE_TRIM = (b"usage: fastq_trim_poly_at --file fastq_file --outfile out_file [optional parameters]\n"
              b"  --help       :print the usage\n"
              b"  --file <filename> :fastq (optional gzipped) file name \n"
              b"  --ofile <filename> : fastq file name where the processed reads will be written \n"
              b"  --min_poly_at_len integer     : minimum length of poly-A|T sequence to remove.\n"
              b"  --min_len integer     : minimum read length.\n")


def trim_poly_at(argv, opener=None, can_write=lambda path: True):
    """-> dict(exit, stdout, stderr, out (decompressed text written to --outfile, or None))"""
    stderr = VERSION
    opts = {"min_poly_at_len": 10, "min_len": 10, "file": None, "outfile": None}
    longs = {"min_poly_at_len": "a", "file": "b", "outfile": "c", "min_len": "d"}
    shorts = {v: k for k, v in longs.items()}
    help_ = False
    i = 0
    while i < len(argv):
        a = argv[i]
        key, val = None, None
        if a == "--":
            break
        if a.startswith("--"):
            name = a[2:]
            if "=" in name:
                name, val = name.split("=", 1)
            cands = [k for k in list(longs) + ["help"] if k.startswith(name)]
            if name in list(longs) + ["help"]:
                cands = [name]
            if len(cands) == 1:
                if cands[0] == "help":
                    help_ = True
                else:
                    key = cands[0]
                    if val is None:
                        i += 1
                        val = argv[i] if i < len(argv) else None
        elif a.startswith("-") and len(a) > 1 and a[1] in shorts:
            key = shorts[a[1]]
            val = a[2:] if len(a) > 2 else None
            if val is None:
                i += 1
                val = argv[i] if i < len(argv) else None
        if key is not None and val is not None:
            if key == "min_poly_at_len":
                v = _atol(val) & 0xFFFFFFFF
                opts[key] = v - (1 << 32) if v & 0x80000000 else v
            elif key == "min_len":
                opts[key] = _atol(val)
            else:
                opts[key] = val
        i += 1
    if help_:
        return {"exit": 0, "stdout": USAGE_TRIM, "stderr": stderr, "out": None}
    stderr += b"INFO:Validating options...\n"
    if opts["file"] is None:
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"missing input file (--file)"), "out": None}
    if opts["outfile"] is None:
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"missing output file name (--outfile)"), "out": None}
    stderr += b"INFO:Options OK.\n"
    try:
        data = read_input(opts["file"], opener)
    except (FileNotFoundError, OSError):
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Unable to open " + opts["file"].encode()), "out": None}
    if not can_write(opts["outfile"]):
        return {"exit": 1, "stdout": b"", "stderr": stderr + err(b"Unable to open " + opts["outfile"].encode()), "out": None}
    records, tail = _lines4(data)
    out, trimmed, discarded = [], 0, 0
    min_len = opts["min_len"] & 0xFFFFFFFFFFFFFFFF
    for h1, seq, h2, qual in records:
        h1, seq, h2, qual = _cstr(h1), _cstr(seq), _cstr(h2), _cstr(qual)
        nseq, nqual, read_len, was = trim_record(seq, qual, opts["min_poly_at_len"])
        if was:
            trimmed += 1
        if read_len >= min_len:
            out += [h1, nseq, h2, nqual if nqual is not None else b""]
        else:
            discarded += 1
    if tail:
        stderr += err(b"Error in file %s: line %d: file truncated" % (opts["file"].encode(), 4 * len(records)))
        return {"exit": 1, "stdout": b"", "stderr": stderr, "out": None}
    stderr += b"INFO:Reads processed: %d\nINFO:Reads trimmed: %d\nINFO:Reads discarded: %d\n" % (
        len(records), trimmed, discarded)
    return {"exit": 0, "stdout": b"", "stderr": stderr, "out": b"".join(out)}
