"""TEST INFRASTRUCTURE, not product code: sequential restatement of the reference's bam_umi_count
(reference src/bam_umi_count.c, 0.25.3, on top of samtools-0.1.19's libbam as vendored in the
reference's deps/) on in-memory BAM images.

Pinned: tests/test_oracle_umi.py requires identical exit status, stderr and output files (.mtx,
_rows, _cols) for every golden invocation in tests/golden/umi_count.json (captured by
tools/gen_golden.py from oracle/_ref/bam_umi_count, the reference program compiled from its own
sources).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pure-Python loops: meant for small inputs.  Each function names the reference lines it follows.
Counters are float32 (numpy) exactly where the reference's are `float`.

The UMI container of a (cell, gene) is the reference's RL_Tree AS IT BEHAVES (oracle/rl_oracle.c,
restating src/range_list.c incl. its defects: members are lost or invented when ids arrive out of
order); `undefined_reads` in the result counts reads of tree memory the reference never wrote, where
its output depends on heap contents (the restatement reads 0 there).
"""
import struct
import zlib

import numpy as np

from oracle.loader import RLTree

VERSION = "0.25.3"
F32 = np.float32
UMIS_FEATURE = 1048576       # src/bam_umi_count.c:48
MAX_CELLS = 1000000          # :43
MAX_FEATURES = 100000        # :44
FEAT_ID_MAX_LEN = 25         # :40
BAM_FUNMAP = 4


class Exit(Exception):
    def __init__(self, status):
        self.status = status


# ---------------------------------------------------------------------------------------------
# BGZF + BAM (SAM/BAM specification; libbam 0.1.19 bam_read1 / bam_aux_get / bam_aux2Z / bam_aux2i)
# ---------------------------------------------------------------------------------------------
def bgzf_inflate(data: bytes) -> bytes:
    out, p = [], 0
    while p + 18 <= len(data):
        if data[p:p + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block")
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        bsize, q = None, p + 12
        while q < p + 12 + xlen:
            si1, si2, slen = data[q], data[q + 1], struct.unpack_from("<H", data, q + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, q + 4)[0]
            q += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        cdata = data[p + 12 + xlen:p + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        p += bsize + 1
    return b"".join(out)


def bam_records(raw: bytes):
    """yield (tid, flag, aux bytes) per alignment; raw = inflated BAM stream"""
    if raw[:4] != b"BAM\x01":
        raise ValueError("not a BAM stream")
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        p += 4 + l_name + 4
    while p + 4 <= len(raw):
        block = struct.unpack_from("<i", raw, p)[0]
        x = struct.unpack_from("<8I", raw, p + 4)
        tid = x[0] - (1 << 32) if x[0] >= (1 << 31) else x[0]
        l_qname, flag, n_cigar, l_qseq = x[2] & 0xFF, x[3] >> 16, x[3] & 0xFFFF, x[4]
        data = raw[p + 36:p + 4 + block]
        aux = data[l_qname + 4 * n_cigar + (l_qseq + 1) // 2 + l_qseq:]
        yield tid, flag, aux
        p += 4 + block


def _type2size(t):  # bam.h:772-778
    if t in b"CcA":
        return 1
    if t in b"Ss":
        return 2
    if t in b"IifF":
        return 4
    return 0


def aux_get(aux: bytes, tag: bytes):
    """bam_aux.c bam_aux_get: offset of the type byte of the first field named `tag`, or None"""
    s, n = 0, len(aux)
    while s < n:
        name = aux[s:s + 2]
        s += 2
        if name == tag[:2]:
            return s
        if s >= n:
            break
        t = bytes([aux[s]]).upper()  # __skip_tag upper-cases the type
        s += 1
        if t in (b"Z", b"H"):
            while s < n and aux[s] != 0:
                s += 1
            s += 1
        elif t == b"B":
            sub = aux[s:s + 1]
            cnt = struct.unpack_from("<i", aux, s + 1)[0] if s + 5 <= n else 0
            s += 5 + _type2size(sub) * cnt
        else:
            s += _type2size(t)
    return None


def aux2Z(aux: bytes, s):
    if s is None:
        return None
    if aux[s:s + 1] in (b"Z", b"H"):
        e = aux.find(b"\0", s + 1)
        return aux[s + 1:e if e >= 0 else len(aux)]
    return None


def aux2i(aux: bytes, s):
    t = aux[s:s + 1]
    if t == b"c":
        return struct.unpack_from("<b", aux, s + 1)[0]
    if t == b"C":
        return aux[s + 1]
    if t == b"s":
        return struct.unpack_from("<h", aux, s + 1)[0]
    if t == b"S":
        return struct.unpack_from("<H", aux, s + 1)[0]
    if t in (b"i", b"I"):
        return struct.unpack_from("<i", aux, s + 1)[0]
    return 0


def get_tag(aux, tag):  # src/bam_umi_count.c:513-522
    z = aux2Z(aux, aux_get(aux, tag))
    return b"" if z is None else z


# ---------------------------------------------------------------------------------------------
# barcodes and label maps
# ---------------------------------------------------------------------------------------------
BASE2INT = {c: v for v, cs in enumerate(["", "Aa", "Cc", "Gg", "Tt", "Nn"]) for c in cs.encode()}
INT2NT = b" ACGTN."


def char2uint_64(s: bytes) -> int:  # :364-382
    if not s:
        return 0
    pos = 0
    while pos < len(s) and s[pos] != 0x0A:
        pos += 1
    if not pos:
        return 0
    i = 0
    pos -= 1
    while pos >= 0:
        b = BASE2INT.get(s[pos], 0)
        if not b:
            break
        i = (i * 10 + b) & 0xFFFFFFFFFFFFFFFF
        pos -= 1
    return i


def uint_642char(i: int) -> bytes:  # :342-360 (digits 7..9 cannot come out of char2uint_64 without overflow)
    out = bytearray()
    while i > 0:
        d = i % 10
        out.append(INT2NT[d] if d < len(INT2NT) else 0x3F)
        i //= 10
    return bytes(out)


class Labels:
    """LABELS / BLABELS (:52-78): dense ids 1..N in order of first appearance"""

    def __init__(self):
        self.ids, self.order = {}, []

    def id_of(self, key):  # label_str2id :143 / blabel2id :225
        i = self.ids.get(key)
        if i is None:
            i = len(self.order) + 1
            self.ids[key] = i
            self.order.append(key)
        return i

    @property
    def ctr(self):
        return len(self.order)


class Feature:
    __slots__ = ("umi", "reads", "ht")

    def __init__(self):
        self.umi, self.reads, self.ht = F32(0), F32(0), None  # ht: RLTree, None = NULL


class Cell:
    __slots__ = ("umi", "reads", "features")

    def __init__(self):
        self.umi, self.reads, self.features = F32(0), F32(0), None  # features: dict feat_id -> Feature


def strtok_tokens(s: bytes):
    """what successive strtok(.., ",") calls return for a C string"""
    return [t for t in s.split(b",") if t]


# ---------------------------------------------------------------------------------------------
# the program
# ---------------------------------------------------------------------------------------------
LONG_FLAGS = {"verbose": ("verbose", 1), "multi_mapped": ("uniq", 0), "uniq_mapped": ("uniq", 1),
              "sorted_by_cell": ("sorted", 1), "not_sorted_by_cell": ("sorted", 0),
              "ignore_sample": ("ignore_sample", 1), "help": ("help", 1), "10x": ("tenx", 1)}
LONG_ARGS = {"bam": "b", "cell_suffix": "s", "known_umi": "k", "known_cells": "c", "ucounts": "u", "rcounts": "r",
             "tag": "x", "cell_tag": "X", "min_reads": "t", "min_umis": "U", "max_cells": "C", "max_feat": "F",
             "feat_cell": "T"}
SHORT_WITH_ARG = set("FTCbUurtxcsX")


def atol(s: str) -> int:
    import re
    m = re.match(r"\s*([+-]?\d+)", s)
    return int(m.group(1)) if m else 0


def parse_args(argv):
    """getopt_long(argc, argv, "F:T:C:b:U:u:r:t:x:c:s:hX:", long_options) as main() uses it (:795-852).
    Only what the golden invocations need: exact long names or unambiguous prefixes, -x style shorts."""
    o = {"sorted": 1, "ignore_sample": 1, "uniq": 0, "help": 0, "tenx": 0, "verbose": 0, "err": []}
    i = 0
    names = list(LONG_FLAGS) + list(LONG_ARGS)
    while i < len(argv):
        a = argv[i]
        i += 1
        if a == "--":
            break
        if a.startswith("--"):
            name, eq, val = a[2:].partition("=")
            cands = [n for n in names if n == name] or [n for n in names if n.startswith(name)]
            if len(cands) != 1:
                o["err"].append("bam_umi_count: unrecognized option '%s'\n" % a if not cands
                                else "bam_umi_count: option '%s' is ambiguous\n" % a)
                continue
            n = cands[0]
            if n in LONG_FLAGS:
                k, v = LONG_FLAGS[n]
                o[k] = v
            else:
                if not eq:
                    if i >= len(argv):
                        o["err"].append("bam_umi_count: option '--%s' requires an argument\n" % n)
                        continue
                    val = argv[i]
                    i += 1
                o[LONG_ARGS[n]] = val
        elif a.startswith("-") and len(a) > 1:
            j = 1
            while j < len(a):
                c = a[j]
                j += 1
                if c == "h":
                    o["help"] = 1
                elif c in SHORT_WITH_ARG:
                    if j < len(a):
                        o[c] = a[j:]
                    elif i < len(argv):
                        o[c] = argv[i]
                        i += 1
                    else:
                        o["err"].append("bam_umi_count: option requires an argument -- '%s'\n" % c)
                    break
                else:
                    o["err"].append("bam_umi_count: invalid option -- '%s'\n" % c)
        # non-option arguments are ignored by the program
    return o


USAGE = ("\nERROR: Usage: bam_umi_count --bam in.bam --ucounts output_filename [--min_reads 0] [--min_umis 0] "
         "[--uniq_mapped|--multi_mapped]  [--dump filename] [--tag gx|tx] [--known_umi file_one_umi_per_line] "
         "[--ucounts_MM |--ucounts_tsv] [--ucounts_MM|--ucounts_tsv] [--ignore_sample] [--cell_suffix suffix] "
         "[--max_cells number] [--max_feat number] [--feat_cell number] [--cell_tag tag] [--sorted_by_cell] [--10x]\n")


def run_bam_umi_count(argv, reader, writable=lambda path: True):
    """argv without the program name.  reader(path) -> bytes or None (missing file).
    Returns {"exit", "stderr", "files": {path: bytes}}."""
    err, files = [], {}
    trees = []

    def finish(status):
        return {"exit": status, "stderr": "".join(err), "files": files,
                "undefined_reads": sum(t.undefined_reads for t in trees),
                "overwrites": sum(t.overwrites for t in trees)}

    try:
        err.append("bam_umi_count version %sb\n" % VERSION)
        o = parse_args(argv)
        err.extend(o["err"])
        if o["help"]:
            err.append(USAGE)
            raise Exit(0)
        if "b" not in o:
            err.append(USAGE)
            raise Exit(1)
        if "u" not in o:
            err.append(USAGE)
            raise Exit(1)
        bam_file, ucounts, rcounts = o["b"], o["u"], o.get("r")
        min_reads, min_umis = atol(o.get("t", "0")) & 0xFFFFFFFF, atol(o.get("U", "0")) & 0xFFFFFFFF
        max_cells = atol(o["C"]) if "C" in o else MAX_CELLS
        max_features = atol(o["F"]) if "F" in o else MAX_FEATURES
        feat_tag = (o["x"][:3] if "x" in o else "GX").encode()
        cell_tag = (o["X"][:3] if "X" in o else "CR").encode()
        umi_tag = b"UB" if o["tenx"] else b"RX"
        sorted_mode, uniq_only = o["sorted"], o["uniq"]
        suffix = o.get("s")
        if sorted_mode:
            max_cells = 1

        feature_map, cells_map, umis_map = Labels(), Labels(), Labels()

        def load_whitelist(path, use_map):  # :543-579
            data = reader(path)
            if data is None:
                err.append("\nERROR: Failed to open file %s\n" % path)
                raise Exit(1)
            err.append("Loading whitelist from %s\n" % path)
            vals, n = set(), 0
            lines = data.split(b"\n")
            if lines and lines[-1] == b"":
                lines.pop()
            for ln in lines:
                # fgets(buf, 200): longer lines arrive in pieces; not needed for the fixtures
                num = char2uint_64(ln + b"\n")
                if use_map is not None:
                    num = use_map.id_of(num)
                vals.add(num)
                n += 1
            err.append("Loading whitelist from %s...done.\n" % path)
            return vals, n

        kumi = kcells = None
        if "k" in o:
            kumi, n = load_whitelist(o["k"], umis_map)
            err.append("UMIs whitelist %d\n" % n)
        if "c" in o:
            kcells, n = load_whitelist(o["c"], None)
            err.append("Cells whitelist %d\n" % n)

        raw = reader(bam_file)
        if raw is None:
            # bam_open fails (bgzf.c reports through perror("open")): PRINT_ERROR + return(PARAMS_ERROR_EXIT_STATUS)
            err.append("open: No such file or directory\n")
            err.append("\nERROR: Failed to open BAM file %s\n" % bam_file)
            raise Exit(1)
        err.append("@min_num_reads=%d\n@min_num_umis=%d\n@uniq mapped reads=%d\n@sorted bam=%d\n@tag=%s\n"
                   "@umi tag=%s\n@unique counts file=%s\n" % (min_reads, min_umis, uniq_only, sorted_mode,
                                                               feat_tag.decode("latin-1"), umi_tag.decode(), ucounts))
        if suffix is not None:
            err.append("@cell_suffix=%s\n" % suffix)
        records = list(bam_records(bgzf_inflate(raw)))
        err.append("Processing %s\n" % bam_file)

        HDR = "%%MatrixMarket matrix coordinate real general\n"
        out_u = out_r = None
        if sorted_mode:
            for path in (ucounts, rcounts):
                if path is None:
                    continue
                if not writable(path):
                    err.append("\nERROR: Failed to open file %s\n" % path)
                    raise Exit(1)
                err.append("Creating MM file %s...\n" % path)
            out_u = []
            out_r = [] if rcounts is not None else None
            err.append("Cells processed\n")

        # state (DB, :101-118); sorted mode keeps ONE cell whose feature slots persist across cells
        cells = {}
        db_reads, db_umi = F32(0), F32(0)
        tot = {"u": 0, "r": 0}
        n_alns = n_tags = n_umis_disc = n_cells_disc = 0
        prev_cell_id = cell_id = 0
        ncells = 0

        def cell2MM(lines, umi_mode, key, cid):  # :666-705
            c = cells.get(1 if sorted_mode else cid)
            if c is None or c.features is None:
                return
            pr = 0
            for cf in range(max_features):
                fe = c.features.get(cf)
                if fe is not None and fe.ht is not None:
                    if fe.reads >= F32(min_reads) and fe.umi >= F32(min_umis):
                        if umi_mode and int(fe.umi) >= 1:
                            lines.append("%d %d %d\n" % (cf, cid, c_round(fe.umi)))
                            tot[key] += int(fe.umi)
                        elif int(fe.reads) >= 1:
                            lines.append("%d %d %d\n" % (cf, cid, c_round(fe.reads)))
                            tot[key] += int(fe.reads)
                    pr += 1
                if F32(pr) >= c.umi:
                    break

        def quick_reset():  # :418-441
            c = cells.get(1)
            if c is None:
                return
            c.umi = c.reads = F32(0)
            if c.features:
                for fe in c.features.values():
                    if fe.umi > 0:
                        fe.ht.all_out()
                        fe.umi = fe.reads = F32(0)

        def process_entry(feat_id, umi_id, cid, incr):  # :444-509
            nonlocal db_reads, db_umi
            if umi_id > UMIS_FEATURE:
                err.append("\nERROR: Too many umi barcodes %d - please rerun and increase the maximum number of umis\n\n" % umi_id)
                raise Exit(1)
            if not sorted_mode and cid > max_cells and max_cells > 1:
                err.append("\nERROR: Too many cells %d - please rerun and increase the cells using the --max_cells parameter\n\n" % cid)
                raise Exit(1)
            if feat_id > max_features:
                err.append("\nERROR: Too many features %d - please rerun and increase the maximum number of features using the --max_feat parameter\n\n" % feat_id)
                raise Exit(1)
            idx = 1 if sorted_mode else cid
            c = cells.get(idx)
            if c is None:
                c = cells[idx] = Cell()
            if c.features is None:
                c.features = {}
            fe = c.features.get(feat_id)
            if fe is None:
                fe = c.features[feat_id] = Feature()
            if fe.ht is None:
                fe.ht = RLTree(UMIS_FEATURE)
                trees.append(fe.ht)
                fe.ht.insert(umi_id)
                fe.umi = F32(fe.umi + incr)
                fe.reads = F32(fe.reads + incr)
                c.reads = F32(c.reads + incr)
                c.umi = F32(c.umi + incr)
                db_reads = F32(db_reads + incr)
                db_umi = F32(db_umi + incr)
                return
            if umi_id not in fe.ht:
                fe.ht.insert(umi_id)
                fe.umi = F32(fe.umi + incr)
                c.umi = F32(c.umi + incr)
                db_umi = F32(db_umi + incr)
            fe.reads = F32(fe.reads + incr)
            c.reads = F32(c.reads + incr)
            db_reads = F32(db_reads + incr)

        for tid, flag, aux in records:  # :942-1060
            n_alns += 1
            if not sorted_mode and n_alns % 100000 == 0:
                err.append("\b" * 15 + "%d" % n_alns)
            if tid < 0 or (flag & BAM_FUNMAP):
                continue
            nh_i = 1
            s = aux_get(aux, b"NH")
            if s is not None:
                nh_i = aux2i(aux, s)
                if nh_i > 1 and uniq_only:
                    continue
            feat = get_tag(aux, feat_tag)
            if not feat:
                continue
            n_tags += 1
            umi = get_tag(aux, umi_tag)
            if not umi:
                continue
            cell = get_tag(aux, cell_tag)
            umi_i = char2uint_64(umi)
            if kumi is not None and umi_i not in kumi:
                n_umis_disc += 1
                continue
            umi_id = umis_map.id_of(umi_i)
            cell_i = char2uint_64(cell)
            if kcells is not None and cell_i not in kcells:
                n_cells_disc += 1
                continue
            cell_id = cells_map.id_of(cell_i)
            if sorted_mode:
                if prev_cell_id != cell_id:
                    if cell_id <= prev_cell_id:
                        err.append("Error: The BAM file does not seem to be sorted by CR\n")
                        # exit(1) flushes what cell2MM has written: the complete cells, behind the header of MM_header
                        # (:708-722) that only the end of main comes back to (:1093-1113)
                        files[ucounts] = HDR + "%-10d %-10d %-15d\n" % (0, 0, 0) + "".join(out_u)
                        if rcounts is not None:
                            files[rcounts] = HDR + "%-10d %-10d %-15d\n" % (0, 0, 0) + "".join(out_r)
                        raise Exit(1)
                    if prev_cell_id != 0:
                        ncells += 1
                        if ncells % 10000 == 0:
                            err.append("\b" * 14 + "%-10d" % ncells)
                        cell2MM(out_u, True, "u", prev_cell_id)
                        if out_r is not None:
                            cell2MM(out_r, False, "r", prev_cell_id)
                        quick_reset()
                prev_cell_id = cell_id
            toks = strtok_tokens(feat)
            n_feat = 0
            for k, f in enumerate(toks):
                if k == 0 or f == toks[k - 1]:
                    n_feat += 1
            if not toks:
                continue
            den = n_feat * nh_i
            with np.errstate(divide="ignore", invalid="ignore"):
                incr = F32(np.float64(1.0) / np.float64(den)) if den != 0 else F32(np.inf)
            f = toks[0]  # the second strtok pass only sees the first token (the commas are gone)
            if len(f) + 1 >= FEAT_ID_MAX_LEN:
                err.append("bam_umi_count: src/bam_umi_count.c:1047: main: Assertion `len1+1 < FEAT_ID_MAX_LEN' failed.\n")
                raise Exit(134)
            process_entry(feature_map.id_of(f), umi_id, cell_id, incr)

        if sorted_mode and cell_id != 0:
            ncells += 1
            if ncells % 10000 == 0:
                err.append("\b" * 14 + "%-10d" % ncells)
            cell2MM(out_u, True, "u", cell_id)
            if out_r is not None:
                cell2MM(out_r, False, "r", cell_id)

        err.append("\b" * 15 + "\n")
        err.append("Alignments processed: %d\n" % n_alns)
        err.append("%s encountered  %d times\n" % (feat_tag.decode("latin-1"), n_tags))
        err.append("%d UMIs discarded\n%d cells discarded\n" % (n_umis_disc, n_cells_disc))
        err.append("%d features\n%d cells\n0 samples\n" % (feature_map.ctr, cells_map.ctr))
        err.append("%f total reads\n%f total UMI\n" % (float(db_reads), float(db_umi)))

        def rows_bytes():
            return "".join("%d\t%s\n" % (i + 1, k.decode("latin-1")) for i, k in enumerate(feature_map.order))

        def cols_bytes():
            sfx = suffix or ""
            return "".join("%d\t%s%s\n" % (i + 1, uint_642char(k).decode("latin-1"), sfx)
                           for i, k in enumerate(cells_map.order))

        if not n_tags:
            if sorted_mode:
                files[ucounts] = HDR + "%-10d %-10d %-15d\n" % (0, 0, 0)
                if rcounts is not None:
                    files[rcounts] = HDR + "%-10d %-10d %-15d\n" % (0, 0, 0)
            err.append("ERROR: no valid alignments tagged with %s were found in %s.\n" % (feat_tag.decode("latin-1"), bam_file))
            raise Exit(1)

        if sorted_mode:
            files[ucounts] = HDR + "%-10d %-10d %-15d\n" % (feature_map.ctr, cells_map.ctr, tot["u"]) + "".join(out_u)
            files[ucounts + "_rows"], files[ucounts + "_cols"] = rows_bytes(), cols_bytes()
            if rcounts is not None:
                files[rcounts] = HDR + "%-10d %-10d %-15d\n" % (feature_map.ctr, cells_map.ctr, tot["r"]) + "".join(out_r)
                files[rcounts + "_rows"], files[rcounts + "_cols"] = rows_bytes(), cols_bytes()
            raise Exit(0)

        def write2MM(path, umi_mode):  # :584-663 (row index = the never-set fe->feat_id = 0)
            if not writable(path):
                err.append("\nERROR: Failed to open file %s\n" % path)
                raise Exit(1)
            err.append("Saving MM file %s...\n" % path)
            files[path + "_rows"], files[path + "_cols"] = rows_bytes(), cols_bytes()
            lines, tot_ctr = [], 0
            for cid in range(0, max_cells):
                c = cells.get(cid)
                if c is None or c.features is None:
                    continue
                pr = 0
                for cf in range(max_features):
                    fe = c.features.get(cf)
                    if fe is not None and fe.ht is not None:
                        if fe.reads >= F32(min_reads) and fe.umi >= F32(min_umis):
                            if umi_mode and int(fe.umi) >= 1:
                                lines.append("0 %d %d\n" % (cid, c_round(fe.umi)))
                                tot_ctr += int(fe.umi)
                            elif int(fe.reads) >= 1:
                                lines.append("0 %d %d\n" % (cid, c_round(fe.reads)))
                                tot_ctr += int(fe.reads)
                        pr += 1
                    if F32(pr) >= c.umi:
                        break
            head = HDR + "%d %d " % (feature_map.ctr, cells_map.ctr)
            if not lines:
                files[path] = head + "%-15d\n" % 0
                err.append("ERROR: 0 quantified features.\n")
                raise Exit(1)
            files[path] = head + "%-15d\n" % len(lines) + "".join(lines)
            err.append("Saving MM file...done.\n#cells/features: %d\n#cells: 0\n#tot expr: %d\n" % (len(lines), tot_ctr))

        write2MM(ucounts, True)
        if rcounts is not None:
            write2MM(rcounts, False)
        raise Exit(0)
    except Exit as e:
        return finish(e.status)


def c_round(x) -> int:
    """(uint)round(float): half away from zero"""
    import math
    v = float(x)
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))
