"""Shared helpers for the tests (no GPU, no product code in here)."""
import gzip
import json
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

_PROGRESS = re.compile("\b{15}[0-9]+")


def read_image(path):
    """The decompressed bytes of a FASTQ file as zlib's gzopen/gzread presents them:
    gzip streams are inflated, anything else is passed through."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    return raw


def load_fastq_info_golden():
    with open(os.path.join(GOLD, "fastq_info.json")) as f:
        return json.load(f)


def strip_progress(s):
    """Drop the PRINT_READS_PROCESSED ticker (src/fastq.h:82): 15 backspaces + a count."""
    return _PROGRESS.sub("", s)
