"""ctypes binding of libfqgpu.so (include/fqg.h).

There is deliberately no fallback: if the library cannot be loaded, or no GPU is present,
every entry point raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfqgpu.so")

MEM_HOST, MEM_DEVICE, MEM_DEVICE_INDEXED = 0, 1, 2
VALIDATE_DEFAULT, VALIDATE_FORCE_EXACT, VALIDATE_NO_STATS, VALIDATE_COUNT_TWICE, VALIDATE_FRAME_ONLY = 0, 1, 2, 4, 8
VALIDATE_TWO_PASS = 16
VALIDATE_NAMES = 32
VALIDATE_INDEX = 128  # the frame will be used: the whole line index in this call, not on demand
VALIDATE_NAME_DIGESTS = 256  # names for a uniqueness-only index: 16-byte digests instead of 64-byte records
NAME_DEFAULT, NAME_CASAVA18, NAME_INTEGER, NAME_UNDEF = 0, 1, 2, -1
SPACE_SEQ, SPACE_COLOUR, SPACE_UNDEF = 0, 1, -1

CODE_NAMES = {
    0: "OK", 1: "TRUNCATED", 2: "WRONG_HEADER", 3: "DUP_NAME", 4: "HDR1_AT", 5: "HDR1_SHORT",
    6: "SEQ_CHAR", 7: "SEQ_UT", 8: "LEN_SMALL", 9: "HDR2_PLUS", 10: "HDR2_DIFF", 11: "QLEN",
    12: "QLEN_CS", 13: "UNPAIRED", 14: "NAME_MISMATCH", 15: "LINE_TOO_LONG", 16: "STOP_NUL",
}

# every symbol include/fqg.h declares (checked by tests/test_abi_symbols.py against the header)
EXPORTS = [
    "fqg_open", "fqg_close", "fqg_last_error", "fqg_abi_version", "fqg_set_stream",
    "fqg_synchronize", "fqg_release_scratch", "fqg_host_alloc", "fqg_host_free", "fqg_probe_readname_format",
    "fqg_probe_space", "fqg_probe_first_record", "fqg_acc_create", "fqg_acc_destroy",
    "fqg_acc_reset", "fqg_acc_read", "fqg_acc_hist_nonzero", "fqg_acc_median", "fqg_acc_export",
    "fqg_acc_merge", "fqg_validate", "fqg_frame_records", "fqg_profile_enable",
    "fqg_profile_reset", "fqg_profile_read", "fqg_synth_record_bytes", "fqg_synth_fastq",
    "fqg_frame_retain", "fqg_frame_release", "fqg_frame_n_records", "fqg_frame_make_current", "fqg_index_create",
    "fqg_index_destroy", "fqg_index_insert_unique", "fqg_index_match_delete", "fqg_index_probe_delete",
    "fqg_index_alive", "fqg_index_n_frames", "fqg_index_frame", "fqg_index_names_captured", "fqg_index_expect_lookups", "fqg_records_gather",
    "fqg_records_gather_output", "fqg_names_compare",
    "fqg_barcodes_transform", "fqg_barcodes_output", "fqg_barcodes_output_begin", "fqg_barcodes_output_wait", "fqg_records_filter", "fqg_records_filter_output",
    "fqg_whitelist_create", "fqg_whitelist_destroy", "fqg_barcodes_whitelist",
    "fqg_census_create", "fqg_census_destroy", "fqg_barcodes_census", "fqg_census_finish", "fqg_census_cells",
    "fqg_census_pairs", "fqg_census_device_pairs",
    "fqg_pack_barcode", "fqg_unpack_barcode", "fqg_bam_index_records", "fqg_bam_add_tags", "fqg_bam_add_tags_output",
    "fqg_umi_count", "fqg_umi_features", "fqg_umi_record_features", "fqg_umi_replayed_features", "fqg_umi_umis",
    "fqg_umi_cells", "fqg_umi_entries", "fqg_umi_emit",
    "fqg_fp_owner", "fqg_names_fingerprints", "fqg_names_fingerprints_acct", "fqg_names_fingerprints_named", "fqg_frame_name_records", "fqg_frame_names_equal", "fqg_device_alloc", "fqg_device_free",
    "fqg_device_copy", "fqg_fpset_create", "fqg_fpset_destroy", "fqg_fpset_insert", "fqg_fpset_insert_named",
    "fqg_fpset_candidates", "fqg_fpset_pair_runs", "fqg_frame_name",
]


class LibraryMissing(RuntimeError):
    pass


class FileState(C.Structure):
    _fields_ = [("is_pe", C.c_int32), ("readname_format", C.c_int32), ("space", C.c_int32),
                ("reserved", C.c_int32)]


class FileStats(C.Structure):
    _fields_ = [("num_rds", C.c_uint64), ("min_rl", C.c_uint64), ("max_rl", C.c_uint64),
                ("min_qual", C.c_uint64), ("max_qual", C.c_uint64)]


class BarcodeParams(C.Structure):
    _fields_ = [("present", C.c_int32 * 6), ("interleaved", C.c_int32 * 2), ("umi_read", C.c_int32),
                ("cell_read", C.c_int32), ("sample_read", C.c_int32), ("phred_encoding", C.c_int32),
                ("min_qual", C.c_int32), ("out_sam", C.c_int32), ("tenx", C.c_int32), ("emit", C.c_int32 * 3),
                ("umi_offset", C.c_int64), ("umi_size", C.c_int64), ("cell_offset", C.c_int64),
                ("cell_size", C.c_int64), ("sample_offset", C.c_int64), ("sample_size", C.c_int64),
                ("read_offset", C.c_int64 * 3), ("read_size", C.c_int64 * 3)]


FILTER_N, FILTER_POLY_AT = 1, 2


class FilterParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("max_n_percent", C.c_uint32), ("min_poly_at_len", C.c_int64),
                ("min_len", C.c_int64)]


class FilterResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_kept", C.c_uint64), ("n_trimmed", C.c_uint64),
                ("n_discarded", C.c_uint64), ("out_bytes", C.c_uint64)]


class BarcodeResult(C.Structure):
    _fields_ = [("n_done", C.c_uint64), ("n_discarded", C.c_uint64), ("n_short", C.c_uint64),
                ("out_bytes", C.c_uint64 * 3), ("iteration", C.c_uint64), ("code", C.c_int32), ("file", C.c_int32)]


READ1, READ2, INDEX1, INDEX2, INDEX3 = 1, 2, 3, 4, 5


class UmiParams(C.Structure):
    _fields_ = [("feat_tag", C.c_char * 2), ("cell_tag", C.c_char * 2), ("umi_tag", C.c_char * 2),
                ("reserved", C.c_char * 2), ("sorted_by_cell", C.c_int32), ("uniq_mapped_only", C.c_int32),
                ("max_cells", C.c_uint32), ("max_features", C.c_uint32), ("min_reads", C.c_uint32),
                ("min_umis", C.c_uint32), ("known_umis", C.POINTER(C.c_uint64)),
                ("known_cells", C.POINTER(C.c_uint64)), ("n_known_umis", C.c_uint64), ("n_known_cells", C.c_uint64),
                ("defer_output", C.c_int32), ("strict_set", C.c_int32),
                ("umi_table_keys", C.POINTER(C.c_uint64)), ("umi_table_ids", C.POINTER(C.c_uint32)),
                ("n_umi_table", C.c_uint64), ("db_start_reads", C.c_float), ("db_start_umi", C.c_float),
                ("db_skip", C.c_uint64)]


class BamTagsParams(C.Structure):
    _fields_ = [("tenx", C.c_int32), ("tx_tag", C.c_int32), ("n_targets", C.c_uint32), ("reserved", C.c_uint32),
                ("tx_off", C.POINTER(C.c_uint32)), ("tx_len", C.POINTER(C.c_uint32)), ("gx_off", C.POINTER(C.c_uint32)),
                ("gx_len", C.POINTER(C.c_uint32)), ("names", C.c_char_p), ("names_bytes", C.c_uint64)]


class BamTagsResult(C.Structure):
    _fields_ = [("n_alignments", C.c_uint64), ("n_tagged", C.c_uint64), ("out_bytes", C.c_uint64), ("record", C.c_uint64),
                ("code", C.c_int32), ("reserved", C.c_int32)]


class UmiResult(C.Structure):
    _fields_ = [("n_alignments", C.c_uint64), ("n_tags_found", C.c_uint64), ("n_umis_discarded", C.c_uint64),
                ("n_cells_discarded", C.c_uint64), ("n_features", C.c_uint64), ("n_cells", C.c_uint64),
                ("n_entries", C.c_uint64 * 2), ("total", C.c_uint64 * 2), ("tot_reads", C.c_float),
                ("tot_umi", C.c_float), ("code", C.c_int32), ("reserved", C.c_int32), ("record", C.c_uint64),
                ("aux", C.c_uint64), ("n_counted", C.c_uint64), ("n_new", C.c_uint64),
                ("unit_increments", C.c_int32), ("reserved3", C.c_int32), ("rl_replayed", C.c_uint64),
                ("rl_changed", C.c_uint64), ("rl_undefined", C.c_uint64), ("rl_unresolved", C.c_uint64)]


class UmiEntry(C.Structure):
    _fields_ = [("row", C.c_uint32), ("col", C.c_uint32), ("value", C.c_uint32)]


class ValidateResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_lines", C.c_uint64), ("consumed", C.c_uint64),
                ("record", C.c_uint64), ("aux0", C.c_uint64), ("aux1", C.c_uint64),
                ("code", C.c_int32), ("stopped", C.c_int32), ("path", C.c_int32),
                ("tail_lines", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Record(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("hdr1_len", C.c_uint32), ("seq_len", C.c_uint32),
                ("hdr2_len", C.c_uint32), ("qual_len", C.c_uint32), ("read_len", C.c_uint32),
                ("reserved", C.c_uint32)]


class IndexResult(C.Structure):
    _fields_ = [("n_entries", C.c_uint64), ("index_mem", C.c_uint64), ("record", C.c_uint64),
                ("code", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  The PyTorch wheel bundles its own libamdhip64.so
    (same SONAME, libamdhip64.so.7, as /opt/rocm's).  If libfqgpu.so pulled in the system copy
    first, a later `import torch` would load a second runtime that sees no GPU.  Loading the
    wheel's copy first makes both sides resolve to it; without torch the system copy is used."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libfqgpu.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
    L.fqg_open.argtypes = [C.c_int, C.POINTER(vp)]
    L.fqg_close.argtypes = [vp]
    L.fqg_close.restype = None
    L.fqg_last_error.argtypes = [vp]
    L.fqg_last_error.restype = C.c_char_p
    L.fqg_set_stream.argtypes = [vp, vp]
    L.fqg_synchronize.argtypes = [vp]
    L.fqg_host_alloc.argtypes = [vp, sz]
    L.fqg_host_alloc.restype = vp
    L.fqg_host_free.argtypes = [vp, vp]
    L.fqg_host_free.restype = None
    L.fqg_probe_readname_format.argtypes = [C.c_char_p]
    L.fqg_probe_space.argtypes = [C.c_char_p]
    L.fqg_probe_first_record.argtypes = [vp, u64, C.c_int, C.POINTER(FileState)]
    L.fqg_acc_create.argtypes = [vp, C.POINTER(vp)]
    L.fqg_acc_destroy.argtypes = [vp]
    L.fqg_acc_destroy.restype = None
    L.fqg_acc_reset.argtypes = [vp]
    L.fqg_acc_read.argtypes = [vp, C.POINTER(FileStats)]
    L.fqg_acc_hist_nonzero.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), sz, C.POINTER(sz)]
    L.fqg_acc_median.argtypes = [vp, vp, C.POINTER(u64)]
    L.fqg_acc_export.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.fqg_acc_merge.argtypes = [vp, vp, sz]
    L.fqg_validate.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, C.POINTER(FileState), C.c_uint32,
                               C.POINTER(ValidateResult)]
    L.fqg_frame_records.argtypes = [vp, u64, u64, vp, C.c_int]
    L.fqg_profile_enable.argtypes = [vp, C.c_int]
    L.fqg_profile_reset.argtypes = [vp]
    L.fqg_profile_read.argtypes = [vp, C.POINTER(KernelTime), sz, C.POINTER(sz)]
    L.fqg_synth_record_bytes.argtypes = [C.c_uint32]
    L.fqg_synth_record_bytes.restype = u64
    L.fqg_synth_fastq.argtypes = [vp, vp, u64, C.c_uint32, u64, u64, C.c_int]
    L.fqg_frame_retain.argtypes = [vp, C.POINTER(vp)]
    L.fqg_frame_release.argtypes = [vp]
    L.fqg_frame_release.restype = None
    L.fqg_frame_n_records.argtypes = [vp]
    L.fqg_frame_n_records.restype = u64
    L.fqg_index_create.argtypes = [vp, u64, C.POINTER(vp)]
    L.fqg_index_destroy.argtypes = [vp]
    L.fqg_index_destroy.restype = None
    L.fqg_index_insert_unique.argtypes = [vp, vp, C.POINTER(FileState), C.POINTER(IndexResult)]
    L.fqg_index_match_delete.argtypes = [vp, vp, C.POINTER(FileState), C.POINTER(IndexResult)]
    L.fqg_index_probe_delete.argtypes = [vp, vp, C.POINTER(FileState), C.POINTER(u64), C.POINTER(IndexResult)]
    L.fqg_index_alive.argtypes = [vp, vp, C.POINTER(C.c_uint8), u64]
    L.fqg_index_expect_lookups.argtypes = [vp, C.c_int]
    L.fqg_index_names_captured.argtypes = [vp]
    L.fqg_index_names_captured.restype = u64
    L.fqg_records_gather.argtypes = [vp, vp, C.POINTER(u64), u64, C.POINTER(u64)]
    L.fqg_records_gather_output.argtypes = [vp, vp, u64]
    L.fqg_index_frame.argtypes = [vp, u64]
    L.fqg_index_frame.restype = vp
    L.fqg_frame_make_current.argtypes = [vp, vp]
    L.fqg_names_compare.argtypes = [vp, vp, C.POINTER(FileState), vp, C.POINTER(FileState), C.POINTER(IndexResult)]
    L.fqg_barcodes_transform.argtypes = [vp, C.POINTER(vp), C.POINTER(FileState), C.POINTER(u64),
                                         C.POINTER(BarcodeParams), u64, u64, C.POINTER(BarcodeResult)]
    L.fqg_barcodes_output.argtypes = [vp, C.c_int, vp, u64]
    L.fqg_census_create.argtypes = [vp, C.POINTER(vp)]
    L.fqg_census_destroy.argtypes = [vp]
    L.fqg_census_destroy.restype = None
    L.fqg_barcodes_census.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(FileState), C.POINTER(u64), C.POINTER(BarcodeParams), u64,
                                      C.POINTER(u64)]
    L.fqg_census_finish.argtypes = [vp, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fqg_census_cells.argtypes = [vp, vp, vp, u64]
    L.fqg_census_pairs.argtypes = [vp, vp, C.POINTER(u64), C.POINTER(u64), u64]
    L.fqg_census_device_pairs.argtypes = [vp, C.c_int]
    L.fqg_census_device_pairs.restype = vp
    L.fqg_barcodes_output_begin.argtypes = [vp, C.c_int, vp, u64]
    L.fqg_barcodes_output_wait.argtypes = [vp]
    L.fqg_records_filter.argtypes = [vp, vp, u64, u64, C.POINTER(FilterParams), C.POINTER(FilterResult)]
    L.fqg_records_filter_output.argtypes = [vp, vp, u64]
    L.fqg_fp_owner.argtypes = [u64, C.c_uint32]
    L.fqg_fp_owner.restype = C.c_uint32
    L.fqg_names_fingerprints.argtypes = [vp, vp, C.POINTER(FileState), u64, C.c_uint32, vp, C.POINTER(u64)]
    L.fqg_names_fingerprints_acct.argtypes = [vp, vp, C.POINTER(FileState), u64, C.c_uint32, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fqg_device_alloc.argtypes = [vp, u64]
    L.fqg_device_alloc.restype = vp  # (a pointer: the ctypes default, c_int, would cut it to 32 bits)
    L.fqg_device_free.argtypes = [vp, vp]
    L.fqg_device_free.restype = None
    L.fqg_device_copy.argtypes = [vp, vp, vp, vp, u64]
    L.fqg_fpset_create.argtypes = [vp, u64, C.POINTER(vp)]
    L.fqg_fpset_destroy.argtypes = [vp]
    L.fqg_fpset_destroy.restype = None
    L.fqg_fpset_insert.argtypes = [vp, vp, vp, u64]
    L.fqg_fpset_insert_named.argtypes = [vp, vp, vp, vp, u64]
    L.fqg_names_fingerprints_named.argtypes = [vp, vp, C.POINTER(FileState), u64, C.c_uint32, vp, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fqg_frame_name_records.argtypes = [vp, vp, C.POINTER(FileState), u64, u64, vp]
    L.fqg_frame_names_equal.argtypes = [vp, vp, C.POINTER(FileState), u64, u64, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fqg_fpset_candidates.argtypes = [vp, vp, C.POINTER(u64), u64, C.POINTER(u64)]
    L.fqg_fpset_pair_runs.argtypes = [vp, vp, C.POINTER(PairSummary), C.POINTER(u64), u64]
    L.fqg_frame_name.argtypes = [vp, vp, C.POINTER(FileState), u64, C.c_char_p, u64]
    L.fqg_frame_name.restype = C.c_int64
    L.fqg_whitelist_create.argtypes = [vp, C.POINTER(u64), u64, C.POINTER(vp)]
    L.fqg_whitelist_destroy.argtypes = [vp]
    L.fqg_whitelist_destroy.restype = None
    L.fqg_barcodes_whitelist.argtypes = [vp, vp, u64, u64, u64, C.c_int64, C.c_int64, vp, C.POINTER(C.c_uint8),
                                         C.POINTER(WhitelistResult)]
    L.fqg_pack_barcode.argtypes = [C.c_char_p]
    L.fqg_pack_barcode.restype = u64
    L.fqg_unpack_barcode.argtypes = [u64, C.c_char_p]
    L.fqg_unpack_barcode.restype = None
    L.fqg_bam_index_records.argtypes = [vp, u64, C.POINTER(u64), u64, C.POINTER(u64), C.POINTER(u64)]
    L.fqg_umi_count.argtypes = [vp, vp, u64, C.c_int, C.POINTER(u64), u64, C.POINTER(UmiParams), C.POINTER(UmiResult)]
    L.fqg_bam_add_tags.argtypes = [vp, vp, u64, C.c_int, C.POINTER(u64), u64, C.POINTER(BamTagsParams), C.POINTER(BamTagsResult)]
    L.fqg_bam_add_tags_output.argtypes = [vp, vp, u64]
    L.fqg_umi_emit.argtypes = [vp, C.POINTER(C.c_uint32), u64, C.c_uint32, C.POINTER(UmiResult)]
    L.fqg_umi_features.argtypes = [vp, vp, u64]
    L.fqg_umi_cells.argtypes = [vp, C.POINTER(u64), u64]
    L.fqg_umi_entries.argtypes = [vp, C.c_int, C.POINTER(UmiEntry), u64]
    _lib = L
    return L


class FqgError(RuntimeError):
    pass


def probe_first_record(image: bytes, is_pe: bool) -> FileState:
    st = FileState()
    st.is_pe = int(is_pe)
    st.readname_format = NAME_UNDEF
    st.space = SPACE_UNDEF
    load().fqg_probe_first_record(image, len(image), int(is_pe), C.byref(st))
    return st


class Accumulator:
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(load().fqg_acc_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            load().fqg_acc_destroy(self.h)
            self.h = None

    def reset(self):
        self.ctx._check(load().fqg_acc_reset(self.h))

    def read(self):
        s = FileStats()
        self.ctx._check(load().fqg_acc_read(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in FileStats._fields_}

    def hist(self):
        n = C.c_size_t()
        self.ctx._check(load().fqg_acc_hist_nonzero(self.h, None, None, 0, C.byref(n)))
        lens = (C.c_uint64 * max(1, n.value))()
        cnts = (C.c_uint64 * max(1, n.value))()
        self.ctx._check(load().fqg_acc_hist_nonzero(self.h, lens, cnts, n.value, C.byref(n)))
        return {int(lens[i]): int(cnts[i]) for i in range(n.value)}

    def median(self, other=None):
        m = C.c_uint64()
        self.ctx._check(load().fqg_acc_median(self.h, other.h if other else None, C.byref(m)))
        return m.value

    def export(self):
        n = C.c_size_t()
        self.ctx._check(load().fqg_acc_export(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self.ctx._check(load().fqg_acc_export(self.h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def merge(self, blob: bytes):
        self.ctx._check(load().fqg_acc_merge(self.h, blob, len(blob)))


class WhitelistResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_valid", C.c_uint64), ("n_short", C.c_uint64)]


class Whitelist:
    """Device set of packed barcodes (fqg_whitelist): what load_whitelist builds from a --known_cells file."""

    def __init__(self, ctx, packed):
        self.ctx = ctx
        arr = (C.c_uint64 * max(1, len(packed)))(*packed)
        h = C.c_void_p()
        ctx._check(load().fqg_whitelist_create(ctx.h, arr, len(packed), C.byref(h)))
        self.h = h

    @classmethod
    def from_lines(cls, ctx, text):
        """the lines of a whitelist file as the reference reads them (fgets + char2uint_64 per non-empty line)"""
        lines = text.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()
        return cls(ctx, [int(load().fqg_pack_barcode(ln + b"\n")) for ln in lines])

    def close(self):
        if self.h:
            load().fqg_whitelist_destroy(self.h)
            self.h = None


class Census:
    """FASTQ -> (cell, UMI) without the BAM round trip (fqg_barcodes_census, include/fqg.h)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(load().fqg_census_create(ctx.h, C.byref(self.h)))

    def finish(self):
        """sorts the pairs by (cell, UMI) and counts; (pairs, cells)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.ctx._check(load().fqg_census_finish(self.ctx.h, self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def cells(self, n_cells):
        """numpy array (n_cells, 3): packed cell, reads, distinct UMIs - ascending packed cell"""
        import numpy as np
        out = np.zeros((max(1, n_cells), 3), dtype=np.uint64)
        self.ctx._check(load().fqg_census_cells(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p), n_cells))
        return out[:n_cells]

    def pairs(self, n_pairs):
        import numpy as np
        ce = np.zeros(max(1, n_pairs), dtype=np.uint64)
        um = np.zeros(max(1, n_pairs), dtype=np.uint64)
        self.ctx._check(load().fqg_census_pairs(self.ctx.h, self.h, ce.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                um.ctypes.data_as(C.POINTER(C.c_uint64)), n_pairs))
        return ce[:n_pairs], um[:n_pairs]

    def device_pairs(self, which):
        return load().fqg_census_device_pairs(self.h, which)

    def close(self):
        if self.h:
            load().fqg_census_destroy(self.h)
            self.h = C.c_void_p()


class Frame:
    """A retained frame (fqg_frame): image + line index that outlive later validate calls."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(load().fqg_frame_retain(ctx.h, C.byref(h)))
        self.h = h

    @property
    def n_records(self):
        return int(load().fqg_frame_n_records(self.h))

    def release(self):
        if self.h:
            load().fqg_frame_release(self.h)
            self.h = None


FP_FILE2 = 1 << 63  # FQG_FP_FILE2: the entry comes from the second file of a pair


class PairSummary(C.Structure):
    _fields_ = [("matched", C.c_uint64), ("leftover", C.c_uint64), ("unpaired", C.c_uint64),
                ("first_unpaired", C.c_uint64), ("n_complex", C.c_uint64)]


NAME_REC_BYTES = 64  # FQG_NAME_REC_BYTES


class FingerprintSet:
    """Owner-side collection of read-name fingerprints (fqg_fpset)."""

    def __init__(self, ctx, expected):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(load().fqg_fpset_create(ctx.h, expected, C.byref(h)))
        self.h = h

    def insert(self, device_ptr, n, names_device_ptr=None):
        """n pairs from device memory; with names_device_ptr also their NAME_REC_BYTES-byte name records (then a
        holder and its asker only pair when the names are the same bytes: fqg_fpset_insert_named)"""
        if names_device_ptr is None:
            self.ctx._check(load().fqg_fpset_insert(self.ctx.h, self.h, C.c_void_p(int(device_ptr)), n))
        else:
            self.ctx._check(load().fqg_fpset_insert_named(self.ctx.h, self.h, C.c_void_p(int(device_ptr)),
                                                          C.c_void_p(int(names_device_ptr)), n))

    def candidates(self, cap=1 << 16):
        pairs = (C.c_uint64 * (2 * cap))()
        found = C.c_uint64()
        self.ctx._check(load().fqg_fpset_candidates(self.ctx.h, self.h, pairs, cap, C.byref(found)))
        k = min(found.value, cap)
        return [(int(pairs[2 * i]), int(pairs[2 * i + 1])) for i in range(k)], int(found.value)

    def pair_runs(self, cap=1 << 16):
        """-> (summary dict, [(run id, index with FP_FILE2 for file 2)] of the runs that need the name bytes)"""
        ent = (C.c_uint64 * (2 * cap))()
        r = PairSummary()
        self.ctx._check(load().fqg_fpset_pair_runs(self.ctx.h, self.h, C.byref(r), ent, cap))
        k = min(r.n_complex, cap)
        out = {f: int(getattr(r, f)) for f, _ in PairSummary._fields_}
        if out["first_unpaired"] == (1 << 64) - 1:
            out["first_unpaired"] = None
        return out, [(int(ent[2 * i]), int(ent[2 * i + 1])) for i in range(k)]

    def close(self):
        if self.h:
            load().fqg_fpset_destroy(self.h)
            self.h = None


class NameIndex:
    """Device read-name index (fqg_index)."""

    def __init__(self, ctx, expected_names=0):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(load().fqg_index_create(ctx.h, expected_names, C.byref(h)))
        self.h = h

    def insert_unique(self, state):
        r = IndexResult()
        self.ctx._check(load().fqg_index_insert_unique(self.ctx.h, self.h, C.byref(state), C.byref(r)))
        return r.as_dict()

    def match_delete(self, state):
        r = IndexResult()
        self.ctx._check(load().fqg_index_match_delete(self.ctx.h, self.h, C.byref(state), C.byref(r)))
        return r.as_dict()

    def probe_delete(self, state, n_records):
        """match_delete with one answer per record of the current frame: the inserted record (insertion order) whose
        entry the record took, or None"""
        r = IndexResult()
        m = (C.c_uint64 * max(1, n_records))()
        self.ctx._check(load().fqg_index_probe_delete(self.ctx.h, self.h, C.byref(state), m, C.byref(r)))
        d = r.as_dict()
        d["match"] = [None if m[i] >= 0xFFFFFFFFFFFFFFFE else int(m[i]) for i in range(n_records)]
        d["wrong_header"] = [i for i in range(n_records) if m[i] == 0xFFFFFFFFFFFFFFFE]
        return d

    def probe_delete_np(self, state, n_records):
        """probe_delete for large frames: (result fields, numpy uint64 array: per record of the current frame the
        inserted record whose entry it took, FQG_NO_MATCH or FQG_MATCH_WRONG_HEADER)"""
        import numpy as np
        r = IndexResult()
        m = np.empty(max(1, n_records), dtype=np.uint64)
        self.ctx._check(load().fqg_index_probe_delete(self.ctx.h, self.h, C.byref(state), m.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      C.byref(r)))
        return r.as_dict(), m[:n_records]

    def alive_np(self, n_inserted):
        import numpy as np
        a = np.empty(max(1, n_inserted), dtype=np.uint8)
        self.ctx._check(load().fqg_index_alive(self.ctx.h, self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), n_inserted))
        return a[:n_inserted]

    def expect_lookups(self, yes):
        """before the first insert: False = the index is only a uniqueness test (it keeps no name records)"""
        self.ctx._check(load().fqg_index_expect_lookups(self.h, 1 if yes else 0))

    def names_captured(self):
        """of the last insert / match call: records whose name came from a capture record of the streaming pass"""
        return int(load().fqg_index_names_captured(self.ctx.h))

    def frame(self, k=0):
        """the k-th frame the index has retained (borrowed)"""
        return load().fqg_index_frame(self.h, k)

    def alive(self, n_inserted):
        a = (C.c_uint8 * max(1, n_inserted))()
        self.ctx._check(load().fqg_index_alive(self.ctx.h, self.h, a, n_inserted))
        return [bool(a[i]) for i in range(n_inserted)]

    def close(self):
        if self.h:
            load().fqg_index_destroy(self.h)
            self.h = None


class Context:
    """One GPU context (fqg_ctx).  Raises if there is no GPU: there is no CPU path."""

    def __init__(self, device=0):
        L = load()
        h = C.c_void_p()
        rc = L.fqg_open(device, C.byref(h))
        if rc != 0:
            raise FqgError(f"fqg_open({device}) failed with {rc}: no MI355X visible?")
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise FqgError(f"libfqgpu error {rc}: {load().fqg_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            load().fqg_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def accumulator(self):
        return Accumulator(self)

    def set_stream(self, stream_ptr):
        self._check(load().fqg_set_stream(self.h, stream_ptr))

    def synchronize(self):
        self._check(load().fqg_synchronize(self.h))

    def validate(self, image, acc, state, final=True, flags=0, nbytes=None, mem=None):
        """image: bytes (host) or an int device pointer (then nbytes is required)."""
        res = ValidateResult()
        if isinstance(image, (bytes, bytearray)):
            buf = (C.c_char * len(image)).from_buffer_copy(image) if len(image) else None
            n = len(image)
            rc = load().fqg_validate(self.h, acc.h if acc else None, buf, n, MEM_HOST, int(final),
                                     C.byref(state), flags, C.byref(res))
        else:
            rc = load().fqg_validate(self.h, acc.h if acc else None, C.c_void_p(int(image)), nbytes,
                                     MEM_DEVICE if mem is None else mem, int(final), C.byref(state),
                                     flags, C.byref(res))
        self._check(rc)
        return res.as_dict()

    def release_scratch(self):
        """give the device buffers the context keeps between calls back (the current frame goes with them)"""
        self._check(load().fqg_release_scratch(self.h))

    def retain_frame(self):
        return Frame(self)

    def barcodes_transform(self, frames, states, n_iterations, umi=None, cell=None, sample=None, phred=33,
                           min_qual=0, sam=True, tenx=False, first_read_number=0):
        """fastq_pre_barcodes' main loop on retained frames.  frames / states: {READ1..INDEX3: Frame / FileState};
        umi / cell / sample: (file reference, offset, size) or None.  Output stays on the device
        (barcodes_output copies it)."""
        p, fr, stt, first = self._barcode_args(frames, states, umi, cell, sample, phred, min_qual, sam, tenx)
        r = BarcodeResult()
        self._check(load().fqg_barcodes_transform(self.h, fr, stt, first, C.byref(p), n_iterations, first_read_number,
                                                  C.byref(r)))
        out = {k: getattr(r, k) for k, _ in BarcodeResult._fields_ if k != "out_bytes"}
        out["out_bytes"] = list(r.out_bytes)
        return out

    def barcodes_census(self, census, frames, states, n_done, umi=None, cell=None, sample=None, phred=33, min_qual=0,
                        sam=True, tenx=False):
        """behind the barcodes_transform of the same batch, with its arguments and its n_done: the (cell, UMI) pairs of the
        reads it kept go to the census (device memory).  Returns how many were added."""
        p, fr, stt, first = self._barcode_args(frames, states, umi, cell, sample, phred, min_qual, sam, tenx)
        added = C.c_uint64(0)
        self._check(load().fqg_barcodes_census(self.h, census.h, fr, stt, first, C.byref(p), n_done, C.byref(added)))
        return int(added.value)

    def census(self):
        return Census(self)

    def _barcode_args(self, frames, states, umi, cell, sample, phred, min_qual, sam, tenx):
        p = BarcodeParams()
        fr = (C.c_void_p * 6)()
        stt = (FileState * 6)()
        first = (C.c_uint64 * 6)()
        for x, f in frames.items():
            p.present[x] = 1
            fr[x] = f.h
            stt[x] = states[x]
        p.umi_read = p.cell_read = p.sample_read = -1
        p.umi_offset = p.cell_offset = p.sample_offset = -1
        p.read_offset[1] = p.read_offset[2] = -1
        for name, spec in (("umi", umi), ("cell", cell), ("sample", sample)):
            if spec:
                setattr(p, name + "_read", spec[0])
                setattr(p, name + "_offset", spec[1])
                setattr(p, name + "_size", spec[2])
        p.phred_encoding, p.min_qual, p.out_sam, p.tenx = phred, min_qual, int(sam), int(tenx)
        if not sam:
            p.emit[1] = 1
            p.emit[2] = 1 if READ2 in frames else 0
        return p, fr, stt, first

    def records_filter(self, frame, n_records, mode, max_n_percent=0, min_poly_at_len=10, min_len=10, first_record=0):
        """fastq_filter_n (mode FILTER_N) / fastq_trim_poly_at (FILTER_POLY_AT) on a retained frame; the kept
        records stay on the device as FASTQ text (records_filter_output copies them)."""
        p = FilterParams(mode, max_n_percent, min_poly_at_len, min_len)
        r = FilterResult()
        self._check(load().fqg_records_filter(self.h, frame.h, first_record, n_records, C.byref(p), C.byref(r)))
        return {k: getattr(r, k) for k, _ in FilterResult._fields_}

    def records_filter_output(self, nbytes):
        buf = C.create_string_buffer(max(1, nbytes))
        self._check(load().fqg_records_filter_output(self.h, buf, nbytes))
        return buf.raw[:nbytes]

    def barcodes_whitelist(self, frame, whitelist, offset, size, n_records=None, first_record=0, step=1, want_flags=False):
        """valid_barcode (src/bam_umi_count.c:523-535) on the `size` characters at `offset` of every record's sequence;
        returns the counts (and one byte per record when want_flags)"""
        import numpy as np
        n = frame.n_records if n_records is None else n_records
        r = WhitelistResult()
        flags = np.zeros(max(1, n), dtype=np.uint8) if want_flags else None
        self._check(load().fqg_barcodes_whitelist(self.h, frame.h, first_record, step, n, offset, size, whitelist.h,
                                                  flags.ctypes.data_as(C.POINTER(C.c_uint8)) if want_flags else None,
                                                  C.byref(r)))
        d = {k: int(getattr(r, k)) for k, _ in WhitelistResult._fields_}
        if want_flags:
            d["valid"] = flags[:n]
        return d

    def barcodes_output(self, which, nbytes, beside_next_call=False):
        """output `which` of the last transform.  beside_next_call: through fqg_barcodes_output_begin / _wait (the copy
        runs on a stream of its own; the caller may launch other work on the context before it waits)"""
        buf = C.create_string_buffer(max(1, nbytes))
        if beside_next_call:
            self._check(load().fqg_barcodes_output_begin(self.h, which, buf, nbytes))
            self._check(load().fqg_barcodes_output_wait(self.h))
        else:
            self._check(load().fqg_barcodes_output(self.h, which, buf, nbytes))
        return buf.raw[:nbytes]

    def names_fingerprints(self, frame, state, record_base, n_owners, out_device_ptr, names_device_ptr=None):
        """(fingerprint, global index) pairs of a retained frame (None: the current one) into device
        memory, bucketed by owner; returns the bucket sizes.  names_device_ptr: also the NAME_REC_BYTES-byte name record
        of every pair, at the same place of that second array."""
        counts = (C.c_uint64 * n_owners)()
        if names_device_ptr is None:
            self._check(load().fqg_names_fingerprints(self.h, frame.h if frame is not None else None, C.byref(state),
                                                      record_base, n_owners, C.c_void_p(int(out_device_ptr)), counts))
        else:
            self._check(load().fqg_names_fingerprints_named(self.h, frame.h if frame is not None else None, C.byref(state),
                                                            record_base, n_owners, C.c_void_p(int(out_device_ptr)),
                                                            C.c_void_p(int(names_device_ptr)), counts, None))
        return [int(x) for x in counts]

    def frame_name_records(self, frame, state, first, n, out_device_ptr):
        """the NAME_REC_BYTES-byte name records of records first .. first + n - 1 of a retained frame into device memory
        (fqg_frame_name_records: names by position, include/fqg.h)"""
        self._check(load().fqg_frame_name_records(self.h, frame.h, C.byref(state), first, n, C.c_void_p(int(out_device_ptr))))

    def frame_names_equal(self, frame, state, first, n, recs_device_ptr):
        """(records of the frame whose name is the name record at the same place, records that agree in the 56 bytes a
        record holds but are longer) - fqg_frame_names_equal"""
        eq, und = C.c_uint64(0), C.c_uint64(0)
        self._check(load().fqg_frame_names_equal(self.h, frame.h, C.byref(state), first, n, C.c_void_p(int(recs_device_ptr)),
                                                 C.byref(eq), C.byref(und)))
        return int(eq.value), int(und.value)

    def fingerprint_set(self, expected):
        return FingerprintSet(self, expected)

    def frame_name(self, frame, state, record):
        buf = C.create_string_buffer(1024)
        n = load().fqg_frame_name(self.h, frame.h, C.byref(state), record, buf, 1024)
        if n < 0:
            self._check(int(n))
        return buf.raw[:n]

    def name_index(self, expected_names=0):
        return NameIndex(self, expected_names)

    def names_compare(self, frame_a, state_a, frame_b=None, state_b=None):
        r = IndexResult()
        self._check(load().fqg_names_compare(self.h, frame_a.h, C.byref(state_a), frame_b.h if frame_b else None,
                                             C.byref(state_b) if state_b is not None else None, C.byref(r)))
        return r.as_dict()

    def records_gather(self, frame_handle, records, want_output=False):
        """fqg_records_gather: the records of a frame (a Frame or a borrowed handle) in the given order (numpy uint64);
        returns the byte count and, with want_output, the text"""
        import numpy as np
        rec = np.ascontiguousarray(records, dtype=np.uint64)
        nb = C.c_uint64()
        h = frame_handle.h if hasattr(frame_handle, "h") else frame_handle
        self._check(load().fqg_records_gather(self.h, h, rec.ctypes.data_as(C.POINTER(C.c_uint64)), rec.size, C.byref(nb)))
        if not want_output:
            return nb.value, None
        dst = C.create_string_buffer(max(1, nb.value))
        self._check(load().fqg_records_gather_output(self.h, dst, nb.value))
        return nb.value, dst.raw[:nb.value]

    def frame_make_current(self, frame_handle):
        h = frame_handle.h if hasattr(frame_handle, "h") else frame_handle
        self._check(load().fqg_frame_make_current(self.h, h))

    def frame_records(self, first, count):
        out = (Record * max(1, count))()
        self._check(load().fqg_frame_records(self.h, first, count, out, MEM_HOST))
        return [{k: getattr(out[i], k) for k, _ in Record._fields_ if k != "reserved"}
                for i in range(count)]

    def profile(self, on=True):
        self._check(load().fqg_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._check(load().fqg_profile_reset(self.h))

    def profile_read(self):
        n = C.c_size_t()
        out = (KernelTime * 64)()
        self._check(load().fqg_profile_read(self.h, out, 64, C.byref(n)))
        return {out[i].name.decode(): (int(out[i].launches), float(out[i].total_ms))
                for i in range(min(64, n.value))}

    def umi_count(self, stream, offsets=None, sorted_by_cell=True, uniq_mapped_only=False, feat_tag=b"GX",
                  cell_tag=b"CR", umi_tag=b"RX", max_cells=None, max_features=100000, min_reads=0, min_umis=0,
                  known_umis=None, known_cells=None, nbytes=None, want_entries=True, defer_output=False,
                  strict_set=False, umi_table=None, db_start=None, db_skip=0, offsets_device=None):
        """bam_umi_count's alignment loop on an inflated BAM stream: bytes (host) or an int device pointer
        (then `nbytes` and `offsets` are required).  Returns the result fields plus, when the call
        succeeded, feature names / packed cells in id order and the (row, col, value) lines."""
        L = load()
        host = isinstance(stream, (bytes, bytearray))
        if host:
            buf = (C.c_char * len(stream)).from_buffer_copy(stream)
            nbytes = len(stream)
        if offsets_device is not None:  # (device pointer, number of records): the index lies in HBM beside the records
            assert not host
            offs, n_rec = C.cast(C.c_void_p(int(offsets_device[0])), C.POINTER(C.c_uint64)), int(offsets_device[1])
        elif offsets is None:
            n, used = C.c_uint64(), C.c_uint64()
            self._check(L.fqg_bam_index_records(buf, nbytes, None, 0, C.byref(n), C.byref(used)))
            offs = (C.c_uint64 * max(1, n.value))()
            self._check(L.fqg_bam_index_records(buf, nbytes, offs, n.value, C.byref(n), C.byref(used)))
            n_rec = n.value
        else:
            n_rec = len(offsets)
            if hasattr(offsets, "c"):       # a wrapper that already holds a ctypes array (bench.py)
                offs = offsets.c
            else:
                offs = offsets if isinstance(offsets, C.Array) else (C.c_uint64 * max(1, n_rec))(*offsets)
        p = UmiParams()
        p.feat_tag, p.cell_tag, p.umi_tag = feat_tag[:2], cell_tag[:2], umi_tag[:2]
        p.sorted_by_cell, p.uniq_mapped_only = int(sorted_by_cell), int(uniq_mapped_only)
        p.max_cells = (1 if sorted_by_cell else 1000000) if max_cells is None else max_cells
        p.max_features, p.min_reads, p.min_umis = max_features, min_reads, min_umis
        p.defer_output = int(defer_output)
        p.strict_set = int(strict_set)
        keep = []
        for name, vals in (("known_umis", known_umis), ("known_cells", known_cells)):
            if vals is not None:
                arr = (C.c_uint64 * max(1, len(vals)))(*vals)
                keep.append(arr)
                setattr(p, name, arr)
                setattr(p, "n_" + name, len(vals))
        if db_start is not None:  # a shard continues the file's float32 chain of totals (fractional increments)
            p.db_start_reads, p.db_start_umi, p.db_skip = float(db_start[0]), float(db_start[1]), int(db_skip)
        if umi_table is not None:  # (sorted packed UMIs, their ids in the whole file): numpy uint64 / uint32 arrays
            tk, ti = umi_table
            keep += [tk, ti]
            p.umi_table_keys = tk.ctypes.data_as(C.POINTER(C.c_uint64))
            p.umi_table_ids = ti.ctypes.data_as(C.POINTER(C.c_uint32))
            p.n_umi_table = len(tk)
        r = UmiResult()
        self._check(L.fqg_umi_count(self.h, buf if host else C.c_void_p(int(stream)), nbytes,
                                    MEM_HOST if host else (MEM_DEVICE_INDEXED if offsets_device is not None else MEM_DEVICE), offs,
                                    n_rec, C.byref(p), C.byref(r)))
        out = self._umi_result(r)
        if r.code == 0 and want_entries:
            names = C.create_string_buffer(max(1, r.n_features * 25))
            self._check(L.fqg_umi_features(self.h, names, r.n_features))
            out["features"] = [names.raw[i * 25:(i + 1) * 25].split(b"\0")[0] for i in range(r.n_features)]
            cells = (C.c_uint64 * max(1, r.n_cells))()
            self._check(L.fqg_umi_cells(self.h, cells, r.n_cells))
            out["cells"] = [int(cells[i]) for i in range(r.n_cells)]
            if not defer_output:
                out["entries"] = self._umi_entries(r)
        return out

    def umi_umis(self):
        """the packed UMIs of the last umi_count that are not whitelisted, in order of first appearance (numpy uint64)"""
        import numpy as np
        L = load()
        L.fqg_umi_umis.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        n = C.c_uint64()
        self._check(L.fqg_umi_umis(self.h, None, 0, C.byref(n)))
        a = np.zeros(max(1, n.value), dtype=np.uint64)
        self._check(L.fqg_umi_umis(self.h, a.ctypes.data, n.value, C.byref(n)))
        return a[:n.value]

    def umi_record_features(self, n_records):
        """after umi_count(defer_output=True): the feature id of every alignment (0: not counted), numpy uint32"""
        import numpy as np
        a = np.zeros(max(1, n_records), dtype=np.uint32)
        L = load()
        L.fqg_umi_record_features.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(L.fqg_umi_record_features(self.h, a.ctypes.data, n_records))
        return a[:n_records]

    def umi_replayed_features(self, n_features):
        """after umi_count(defer_output=True): per feature id (index 0 unused) whether one of its sets was replayed"""
        import numpy as np
        a = np.zeros(n_features + 1, dtype=np.uint8)
        L = load()
        L.fqg_umi_replayed_features.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(L.fqg_umi_replayed_features(self.h, a.ctypes.data, n_features + 1))
        return a

    def bam_add_tags(self, stream, tenx=False, tx_tag=False, targets=(), genes=None, offsets=None, nbytes=None,
                     want_output=True):
        """bam_add_tags' alignment loop (src/bam_add_tags.c:250-294) on an inflated BAM stream: bytes (host) or an
        int device pointer (then `nbytes` and `offsets` are required).  targets: the header's reference names;
        genes: {transcript name: gene} (--tx_2_gx) or None.  Returns the result fields and, with want_output, the
        records with their new tags (everything behind the header) as bytes."""
        L = load()
        host = isinstance(stream, (bytes, bytearray))
        if host:
            buf = (C.c_char * max(1, len(stream))).from_buffer_copy(stream)
            nbytes = len(stream)
        if offsets is None:
            n, used = C.c_uint64(), C.c_uint64()
            self._check(L.fqg_bam_index_records(buf, nbytes, None, 0, C.byref(n), C.byref(used)))
            offs = (C.c_uint64 * max(1, n.value))()
            self._check(L.fqg_bam_index_records(buf, nbytes, offs, n.value, C.byref(n), C.byref(used)))
            n_rec = n.value
        else:
            n_rec = len(offsets)
            offs = offsets.c if hasattr(offsets, "c") else (offsets if isinstance(offsets, C.Array) else (C.c_uint64 * max(1, n_rec))(*offsets))
        nt = len(targets)
        blob, tx_off, tx_len, gx_off, gx_len = bytearray(), [], [], [], []
        for t in targets:
            tx_off.append(len(blob)); tx_len.append(len(t)); blob += t + b"\0"
            g = genes.get(t) if genes is not None else None
            if g is None:
                gx_off.append(0); gx_len.append(0xFFFFFFFF)
            else:
                gx_off.append(len(blob)); gx_len.append(len(g)); blob += g + b"\0"
        arr = lambda v: (C.c_uint32 * max(1, nt))(*v)
        p = BamTagsParams()
        p.tenx, p.tx_tag, p.n_targets = int(tenx), int(tx_tag), nt
        keep = [arr(tx_off), arr(tx_len), arr(gx_off), arr(gx_len), bytes(blob) + b"\0"]
        p.tx_off, p.tx_len, p.gx_off, p.gx_len = keep[0], keep[1], keep[2], keep[3]
        p.names, p.names_bytes = keep[4], len(blob)
        r = BamTagsResult()
        self._check(L.fqg_bam_add_tags(self.h, buf if host else C.c_void_p(int(stream)), nbytes,
                                       MEM_HOST if host else MEM_DEVICE, offs,
                                       n_rec, C.byref(p), C.byref(r)))
        out = {k: getattr(r, k) for k, _ in BamTagsResult._fields_ if k != "reserved"}
        if r.code == 0 and want_output:
            dst = C.create_string_buffer(max(1, r.out_bytes))
            self._check(L.fqg_bam_add_tags_output(self.h, dst, r.out_bytes))
            out["records"] = dst.raw[:r.out_bytes]
        return out

    @staticmethod
    def _umi_result(r):
        out = {k: getattr(r, k) for k, _ in UmiResult._fields_ if not k.startswith("reserved") and k not in ("n_entries", "total")}
        out["n_entries"], out["total"] = list(r.n_entries), list(r.total)
        return out

    def _umi_entries(self, r):
        both = []
        for w in range(2):
            e = (UmiEntry * max(1, r.n_entries[w]))()
            self._check(load().fqg_umi_entries(self.h, w, e, r.n_entries[w]))
            both.append([(e[i].row, e[i].col, e[i].value) for i in range(r.n_entries[w])])
        return both

    def umi_emit(self, feat_remap=None, cell_offset=0, want_entries=True):
        """The output step after umi_count(defer_output=True): feat_remap[local id] = id to use (index 0 unused)."""
        r = UmiResult()
        if feat_remap is not None:
            arr = (C.c_uint32 * len(feat_remap))(*feat_remap)
            self._check(load().fqg_umi_emit(self.h, arr, len(feat_remap), cell_offset, C.byref(r)))
        else:
            self._check(load().fqg_umi_emit(self.h, None, 0, cell_offset, C.byref(r)))
        out = {"n_entries": list(r.n_entries), "total": list(r.total)}
        if want_entries:
            out["entries"] = self._umi_entries(r)
        return out

    def synth_fastq(self, device_ptr, n_records, read_len=150, first_index=0, seed=12345, mate=1):
        self._check(load().fqg_synth_fastq(self.h, C.c_void_p(int(device_ptr)), n_records, read_len,
                                           first_index, seed, mate))


def synth_record_bytes(read_len=150):
    return int(load().fqg_synth_record_bytes(read_len))
