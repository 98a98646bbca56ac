"""The PMC traffic file bench.py quotes must belong to the kernel sources in the tree: bench.py refuses a stale
file (traffic null), so a change to the framing kernels without a new `tools/profile_round.sh <tag> traffic`
collection shows up here, on CPU, rather than as a missing `roofline.traffic` on the GPU box."""
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("fqg_bench_module", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_traffic_file_matches_the_kernel_sources():
    b = _bench()
    gb, src = b.committed_traffic("k_stream_pass1", 100_000_000, 150, 349)
    assert src is not None and not src.startswith("stale"), src
    assert gb is not None
    # pass 1 reads the image once: traffic within 10 % above the algorithmic bytes, never below them
    assert 34.9 <= gb <= 34.9 * 1.10, gb
    with open(os.path.join(REPO, src)) as f:
        assert json.load(f)["kernel_sources_digest"] == b.kernel_sources_digest()


def test_stale_traffic_file_is_refused(tmp_path, monkeypatch):
    b = _bench()
    real = b.kernel_sources_digest()
    monkeypatch.setattr(b, "kernel_sources_digest", lambda: "0" * 16)
    gb, src = b.committed_traffic("k_stream_pass1", 100_000_000, 150, 349)
    assert gb is None and src.startswith("stale")
    assert real != "0" * 16


def test_other_workload_has_no_traffic_figure():
    b = _bench()
    assert b.committed_traffic("k_stream_pass1", 7_000_000, 150, 349) == (None, None)
