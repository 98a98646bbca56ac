"""TEST INFRASTRUCTURE: ctypes bindings for oracle/liboracle_fq.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_fq.so")
REF_DIR = os.path.join(HERE, "_ref")

FLAG_R, FLAG_S, FLAG_E, FLAG_Q = 1, 2, 4, 8
ARG2_NONE, ARG2_FILE, ARG2_PE = 0, 1, 2


class Job(C.Structure):
    _fields_ = [
        ("buf1", C.c_char_p), ("n1", C.c_size_t), ("name1", C.c_char_p),
        ("buf2", C.c_char_p), ("n2", C.c_size_t), ("name2", C.c_char_p),
        ("arg2_kind", C.c_int), ("flags", C.c_int),
    ]


class Outcome(C.Structure):
    _fields_ = [
        ("code", C.c_int32), ("file", C.c_int32), ("record", C.c_uint64), ("line", C.c_uint64),
        ("aux0", C.c_uint64), ("aux1", C.c_uint64),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("num_reads", C.c_uint64), ("min_rl", C.c_uint64), ("max_rl", C.c_uint64),
        ("median_rl", C.c_uint64), ("min_qual", C.c_uint64), ("max_qual", C.c_uint64),
        ("num_rds_counted", C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("exit_status", C.c_int), ("first", Outcome), ("summary", Summary),
        ("out", C.c_void_p), ("out_len", C.c_size_t), ("err", C.c_void_p), ("err_len", C.c_size_t),
    ]


_lib = None


def build():
    """(Re)build the restatement; also rebuilds oracle/_ref when the reference is present."""
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.fqo_fastq_info.argtypes = [C.POINTER(Job), C.POINTER(Result)]
        _lib.fqo_fastq_info.restype = C.c_int
        _lib.fqo_result_free.argtypes = [C.POINTER(Result)]
        _lib.fqo_qual_range_to_enc.argtypes = [C.c_uint, C.c_uint]
        _lib.fqo_qual_range_to_enc.restype = C.c_char_p
        # rl_oracle.c: the reference's RL_Tree as it behaves
        _lib.orl_new.restype = C.c_void_p
        _lib.orl_new.argtypes = [C.c_uint64]
        _lib.orl_free.argtypes = [C.c_void_p]
        _lib.orl_insert.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orl_member.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orl_member.restype = C.c_int
        _lib.orl_all_out.argtypes = [C.c_void_p]
        for f in ("orl_size", "orl_undefined_reads", "orl_wild_writes", "orl_overwrites"):
            getattr(_lib, f).argtypes = [C.c_void_p]
            getattr(_lib, f).restype = C.c_uint64
        _lib.orl_node.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orl_node.restype = C.c_uint16
        _lib.orl_ever_written.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orl_first_difference.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orl_first_difference.restype = C.c_long
        _lib.orl_replay.argtypes = [C.c_uint64] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p]
    return _lib


class RLTree:
    """The reference's RL_Tree (src/range_list.c) as restated in oracle/rl_oracle.c."""

    def __init__(self, max_size=1048576):
        self.L = lib()
        self.h = self.L.orl_new(max_size)

    def insert(self, number):  # set_in_rl(tree, number, IN)
        self.L.orl_insert(self.h, number)

    def __contains__(self, number):  # in_rl
        return bool(self.L.orl_member(self.h, number))

    def all_out(self):  # rl_all(tree, OUT)
        self.L.orl_all_out(self.h)

    @property
    def size(self):
        return self.L.orl_size(self.h)

    @property
    def undefined_reads(self):
        return self.L.orl_undefined_reads(self.h)

    @property
    def overwrites(self):
        return self.L.orl_overwrites(self.h)

    @property
    def wild_writes(self):
        return self.L.orl_wild_writes(self.h)

    def node(self, idx):
        return self.L.orl_node(self.h, idx)

    def first_difference(self, other_nodes):
        """first live slot (written at some time) that differs from another implementation's uint16 array, or -1"""
        return self.L.orl_first_difference(self.h, other_nodes)

    def ever_written(self, idx):
        return bool(self.L.orl_ever_written(self.h, idx))

    def __del__(self):
        try:
            self.L.orl_free(self.h)
        except Exception:
            pass


def rl_replay(tree_of, umi, epoch, incr, n_trees):
    """is_new per record + (undefined reads, wild writes, overwrites): oracle/rl_oracle.c orl_replay."""
    import numpy as np
    tree_of = np.ascontiguousarray(tree_of, dtype=np.uint32)
    umi = np.ascontiguousarray(umi, dtype=np.uint32)
    epoch = np.ascontiguousarray(epoch, dtype=np.uint32)
    incr = np.ascontiguousarray(incr, dtype=np.float32)
    n = tree_of.size
    is_new = np.zeros(n, dtype=np.uint8)
    stats = np.zeros(3, dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().orl_replay(n, P(tree_of), P(umi), P(epoch), P(incr), int(n_trees), P(is_new), P(stats))
    return is_new, tuple(int(x) for x in stats)


def fastq_filterpair(buf1, name1, buf2, name2, sorted_mode=False):
    """The restated fastq_filterpair (oracle/fq_oracle.c) on two decompressed images.
    Returns dict(exit, stderr, stdout, files=[paired1, paired2, unpaired] uncompressed)."""
    L = lib()
    L.fqo_fastq_filterpair.argtypes = [C.POINTER(Job), C.POINTER(Result), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    job = Job(buf1, len(buf1), name1.encode(), buf2, len(buf2), name2.encode(), ARG2_FILE, FLAG_S if sorted_mode else 0)
    res = Result()
    outs = (C.c_void_p * 3)()
    lens = (C.c_size_t * 3)()
    L.fqo_fastq_filterpair(C.byref(job), C.byref(res), outs, lens)
    d = {"exit": res.exit_status, "stdout": C.string_at(res.out, res.out_len).decode("latin-1"),
         "stderr": C.string_at(res.err, res.err_len).decode("latin-1"),
         "files": [C.string_at(outs[k], lens[k]) for k in range(3)]}
    L.fqo_result_free(C.byref(res))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for k in range(3):
        libc.free(outs[k])
    return d


def parse_args(args):
    """Split a fastq_info argv (without argv[0]) the way the reference does
    (src/fastq_info.c:214-267): flags first, then one or two positionals."""
    flags, pos = 0, []
    for a in args:
        if a.startswith("-") and len(a) > 1 and not pos:
            for ch in a[1:]:
                flags |= {"r": FLAG_R, "s": FLAG_S, "e": FLAG_E, "q": FLAG_Q}[ch]
        else:
            pos.append(a)
    return flags, pos


def fastq_info(buf1, name1, buf2=None, name2=None, arg2_kind=ARG2_NONE, flags=0):
    """Run the restated fastq_info on in-memory (decompressed) file images.
    Returns dict(exit, stdout, stderr, first=dict(...), summary=dict(...))."""
    L = lib()
    job = Job(buf1, len(buf1), name1.encode(), buf2, len(buf2) if buf2 is not None else 0,
              name2.encode() if name2 is not None else None, arg2_kind, flags)
    res = Result()
    L.fqo_fastq_info(C.byref(job), C.byref(res))
    out = C.string_at(res.out, res.out_len)
    err = C.string_at(res.err, res.err_len)
    d = {
        "exit": res.exit_status,
        "stdout": out.decode("latin-1"),
        "stderr": err.decode("latin-1"),
        "first": {k: getattr(res.first, k) for k, _ in Outcome._fields_},
        "summary": {k: getattr(res.summary, k) for k, _ in Summary._fields_},
    }
    L.fqo_result_free(C.byref(res))
    return d
