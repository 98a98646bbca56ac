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

    def __init__(self, fn, keys, workers=12, select=None):
        # select(keys) -> the keys that are run on this box (asked when the first key is: `thinned` below measures the
        # box); a test whose key is not among them is skipped, and says so
        self.fn, self.keys, self.workers, self.results, self.select = fn, list(keys), workers, None, select

    def _safe(self, key):
        try:
            return self.fn(key)
        except BaseException as e:  # (handed to the test of this key)
            return e

    def get(self, key):
        if self.results is None:
            from concurrent.futures import ThreadPoolExecutor

            if self.select is not None:
                chosen = set(self.select(self.keys))
                self.keys = [k for k in self.keys if k in chosen]
            with ThreadPoolExecutor(self.workers) as ex:
                self.results = list(ex.map(self._safe, self.keys))
        if key not in self.keys:
            import pytest
            pytest.skip("a box that starts programs slowly: this repeated case is left out (FQGPU_TEST_THIN=1 runs all)")
        r = self.results[self.keys.index(key)]
        if isinstance(r, BaseException):
            raise r
        return r


# ---- a time budget for the program sweeps ---------------------------------------------------------------------------
# The GPU suite is bounded by program starts (a start is 0.3 s of runtime initialisation, and a box admits so many a
# second however many wait: profiles/r05a_golden_concurrency.txt - 15 on the builder's box; the driver's ran round 4's
# suite 2.5 times slower).  The sweeps that run EVERY golden invocation again under another configuration measure the
# box first - 24 starts, side by side - and on a slow one take every second or third invocation of their list (chosen
# by a hash of the arguments, so the same ones every time; the sweep of the default configuration always runs them
# all).  Since round 6 the thinning is OPT-IN (FQGPU_TEST_THIN=auto measures the box, =2 / =3 force it): by default every
# sweep runs every invocation, whatever the box.
_THIN = None


def start_rate():
    """program starts per second on this box, 12 at a time (bin/fastq_info -r on a four-line file)"""
    import subprocess
    import tempfile
    import time
    from concurrent.futures import ThreadPoolExecutor
    exe = os.path.join(REPO, "bin", "fastq_info")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "t.fastq"), "wb") as f:
            f.write(b"@r\nACGT\n+\nIIII\n")

        def once(_):
            return subprocess.run(["fastq_info", "-r", "t.fastq"], executable=exe, cwd=d, capture_output=True, timeout=120).returncode

        once(0)  # (the first start of a box pages the runtime in)
        t0 = time.time()
        with ThreadPoolExecutor(12) as ex:
            codes = list(ex.map(once, range(24)))
        dt = time.time() - t0
    return 24.0 / dt if all(c == 0 for c in codes) and dt > 0 else 0.0


def sweep_thinning():
    """1: every invocation (the default since round 6: a box that runs less than the builder's is no longer decided
    silently); FQGPU_TEST_THIN=2 / 3: every second / third one; =auto: measure the box's program starts and thin below 10
    a second, as rounds 4 - 5 did by themselves"""
    global _THIN
    if _THIN is None:
        forced = os.environ.get("FQGPU_TEST_THIN", "")
        if forced == "auto":
            rate = start_rate()
            _THIN = 1 if rate >= 10.0 or rate == 0.0 else 2 if rate >= 5.0 else 3
            if _THIN > 1:
                import warnings
                warnings.warn(f"this box starts {rate:.1f} programs a second: the repeated golden sweeps take every "
                              f"{'second' if _THIN == 2 else 'third'} invocation (FQGPU_TEST_THIN=auto)")
        else:
            _THIN = max(1, int(forced)) if forced else 1
    return _THIN


def thinned(cases, key=lambda c: " ".join(c["args"])):
    """the cases a repeated sweep runs on this box (all of them on a box that starts 10 programs a second or more)"""
    import zlib
    t = sweep_thinning()
    if t <= 1:
        return list(cases)
    return [c for c in cases if zlib.crc32(key(c).encode()) % t == 0]
