"""CPU-only checks of the boundary: libfqgpu.so loads without a GPU, exports every function that
include/fqg.h declares, and refuses to open a context when there is no device (no CPU path)."""
import ctypes
import os
import re

import pytest

import fastq_utils_amd as fq
from tests.util import REPO


def declared_functions():
    text = open(os.path.join(REPO, "include", "fqg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fqg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(fq.abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = fq.abi.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.fqg_abi_version() == 1


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(fq.abi.Record) == 32
    assert ctypes.sizeof(fq.abi.ValidateResult) == 64
    assert ctypes.sizeof(fq.abi.FileState) == 16
    assert ctypes.sizeof(fq.abi.IndexResult) == 32


def test_probes_follow_the_reference_ladder():
    L = fq.abi.load()
    assert L.fqg_probe_readname_format(b"A80910ABXX:2:1:20677:2129 1:N:0:ACGT\n") == fq.abi.NAME_CASAVA18
    assert L.fqg_probe_readname_format(b"12345\n") == fq.abi.NAME_INTEGER
    assert L.fqg_probe_readname_format(b"read_without_suffix\n") == fq.abi.NAME_INTEGER  # NOP == 2
    assert L.fqg_probe_readname_format(b"read/1\n") == fq.abi.NAME_DEFAULT
    assert L.fqg_probe_space(b"T0123012301\n") == fq.abi.SPACE_COLOUR
    assert L.fqg_probe_space(b"ACGTACGT\n") == fq.abi.SPACE_SEQ
    st = fq.abi.probe_first_record(b"@r/1\nACGT\n+\nIIII\n", True)
    assert (st.readname_format, st.space, st.is_pe) == (fq.abi.NAME_DEFAULT, fq.abi.SPACE_SEQ, 1)


def test_no_gpu_means_no_context():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fq.abi.FqgError):
        fq.Context(0)
