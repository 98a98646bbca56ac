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


def free_port():
    """A TCP port nobody listens on right now (for torch.distributed rendezvous: fixed ports collide between
    concurrent runs and with sockets in TIME_WAIT)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_group(cmd, timeout, **kw):
    """subprocess.run(capture_output=True) for a launcher that starts ranks of its own: the launcher runs in its
    own process group, and on a timeout the WHOLE group is killed (a hung rank must not outlive the test)."""
    import signal
    import subprocess

    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        raise AssertionError("timed out after %d s: %s\n%s" % (timeout, " ".join(map(str, cmd)), err.decode("latin-1")[-2000:]))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


class SideBySide:
    """What `fn(key)` gives for every key of `keys`, all of them started the first time ONE of them is asked for and
    run side by side (`workers` at a time).  For parametrized tests that each start a program: a process start + HIP
    initialisation is 0.34 s alone, while the driver admits about 15 a second when many wait
    (tools/golden_concurrency.py) - every test still compares its own case, it only does not wait alone.
    An exception inside fn(key) is raised in the test that asks for that key."""

    def __init__(self, fn, keys, workers=12):
        self.fn, self.keys, self.workers, self.results = fn, list(keys), workers, None

    def _safe(self, key):
        try:
            return self.fn(key)
        except BaseException as e:  # (handed to the test of this key)
            return e

    def get(self, key):
        if self.results is None:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(self.workers) as ex:
                self.results = list(ex.map(self._safe, self.keys))
        r = self.results[self.keys.index(key)]
        if isinstance(r, BaseException):
            raise r
        return r
