#!/usr/bin/env python3
"""bench.py - Mreads/s validated by the fastq_info hot path on MI355X.

A "step" is one pass of the hot path (framing + validation + statistics, i.e. what
`fastq_info -r` does per file) over one HBM-resident batch of synthetic 150 bp reads
(BASELINE.json configs[1]: 100 M reads, 349 B/record = 34.9 GB per GPU).  Input bytes are
already in HBM when the timed region starts.  With --gpus N every rank validates its own shard
of the same size (weak scaling; no data-path collective - records are independent), and the
per-rank statistics are merged once at the end.

The LAST stdout line of rank 0 is the bench line of the driver contract: compact (<= 4 KB), plain ASCII, with
  roofline      HBM roofline of the dominant kernel, measured with hipEvents on the launch stream
  cpu_baseline  the reference's own fastq_info -r (oracle/_ref) timed on this box's host cores
  host_fed      the configs[1] placement (reads in host RAM, fed over PCIe): C-ABI and program rates
  extras        a few numbers per untimed extra; extras_ok = every comparison in them held
The same line (marked "final": false, without host_fed/extras) is printed once before the untimed extras start, so a
crash in an extra cannot take the measured numbers with it; each extra then goes out as ONE line of its own
({"extra": name, ...}), and all of them are written to --extras-out as one JSON document.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_READ_EXTRA = 32  # descriptor bytes per record

HEADLINE_MAX_BYTES = 4096  # the driver keeps a bounded tail of stdout: the last line must fit in it with room to spare


def _plain(v, sig=6):
    """JSON-ready copy of v: floats rounded to `sig` significant digits, strings reduced to printable ASCII (a captured
    stderr once carried a run of backspaces into the line), numpy/torch scalars to Python numbers."""
    if isinstance(v, dict):
        return {str(k): _plain(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x, sig) for x in v]
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    if isinstance(v, bytes):
        v = v.decode("latin-1")
    if isinstance(v, str):
        return "".join(c if 32 <= ord(c) < 127 else " " for c in v)
    if hasattr(v, "item"):
        return _plain(v.item(), sig)
    return _plain(str(v), sig)


def _all_checks_hold(block):
    """every boolean of an extra whose key says it is a comparison (ok / *identical* / as_expected / ...) is true and
    no "error" key is anywhere in it"""
    if isinstance(block, dict):
        for k, x in block.items():
            if k == "error":
                return False
            if isinstance(x, bool) and (k == "ok" or k.endswith("_ok") or "identical" in k or k.startswith("as_expected")
                                        or k.endswith("as_expected") or k == "properties_hold"):
                if not x:
                    return False
            if not _all_checks_hold(x):
                return False
    elif isinstance(block, (list, tuple)):
        return all(_all_checks_hold(x) for x in block)
    return True


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def extras_summary(extras):
    """a few numbers per extra for the bench line (the full blocks are the {"extra": ...} lines and --extras-out)"""
    e = extras
    s = {
        "default_mode_ms_over_r": _get(e, "default_mode_extra", "ms_over_validate_only"),
        "name_insert_ms": _get(e, "default_mode_extra", "insert_ms"),
        "file2_match_ms": _get(e, "default_mode_extra", "file2_loop", "match_ms"),
        "pre_barcodes_200M_kernels_ms": _get(e, "pre_barcodes_extra", "kernels_ms"),
        "pre_barcodes_GBps": _get(e, "pre_barcodes_extra", "achieved_GBps_kernels"),
        "pre_barcodes_generic_50M_ms": _get(e, "pre_barcodes_extra", "generic_file_set", "kernels_ms"),
        "census_200M_ms": _get(e, "pre_barcodes_extra", "census_stage", "census_kernel_ms"),
        "pre_barcodes_program_200M_sam_s": (lambda v: min(v) if v else None)(_get(e, "pre_barcodes_extra", "programs", "legs", "sam_to_stdout", "seconds")),
        "fastq_info_r_one_context_Mreads_per_s": _get(e, "e2e", "cli_fastq_info_r_tmpfs_file", "variants", "one_context", "Mreads_per_s"),
        "filter_n_ms": _get(e, "filters_extra", "filter_n", "kernels_ms"),
        "trim_poly_at_ms": _get(e, "filters_extra", "trim_poly_at", "kernels_ms"),
        "umi_count_kernels_ms": _get(e, "umi_count_extra", "kernels_ms"),
        "umi_count_call_ms": _get(e, "umi_count_extra", "wall_ms_one_call_incl_allocations"),
        "umi_count_call_ms_index_in_hbm": _get(e, "umi_count_extra", "wall_ms_one_call_index_in_hbm"),
        "umi_matrix_identical_to_reference": _get(e, "umi_count_extra", "matrix_identical_to_reference_program"),
        "filterpair_kernels_ms": _get(e, "filterpair_extra", "kernels_ms"),
        "bam_add_tags_kernels_ms": _get(e, "bam_add_tags_extra", "kernels_ms"),
        "short_reads_GBps": _get(e, "read_shapes_extra", "short_reads_30_150bp", "GBps_wall"),
        "long_reads_GBps": _get(e, "read_shapes_extra", "long_reads_2_20kb", "GBps_wall"),
    }
    return {k: v for k, v in s.items() if v is not None}


def headline_line(out, extras=None, final=True):
    """The bench line: the contract's keys + roofline + cpu_baseline + host_fed + a short extras summary, serialised
    as ONE line of printable ASCII no longer than HEADLINE_MAX_BYTES.  Anything that would push it over the limit is
    dropped from the optional end (extras, then host_fed details), never from the contract's keys."""
    keep_roof = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "traffic_source",
                 "frac_on_measured_bytes", "step_frac_wall", "algorithmic_bytes_per_step", "dominant_ms_per_step",
                 "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "all_kernels_ms_per_step",
                 "pipeline_achieved", "pipeline_frac", "launches_per_step", "kernels_ms_per_step")
    keep_cpu = ("value", "unit", "cores", "kind", "sample", "seconds", "ok")
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config") if k in out}
    line["roofline"] = {k: out["roofline"][k] for k in keep_roof if k in out["roofline"]}
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: cb[k] for k in keep_cpu if k in cb}
        ac = cb.get("all_cores")
        if ac:
            line["cpu_baseline"]["all_cores"] = {k: ac[k] for k in ("value", "unit", "cores", "kind", "seconds", "ok") if k in ac}
    if out.get("host_fed") is not None:
        line["host_fed"] = out["host_fed"]
    if out.get("dedup_extra") is not None:
        d = out["dedup_extra"]
        line["dedup"] = {k: d[k] for k in ("error", "names_total", "finding", "wall_ms_max_over_ranks",
                                           "Mnames_per_s_whole_job") if k in d}
        if isinstance(d.get("pairing"), dict):
            line["dedup"]["pairing_ok"] = d["pairing"].get("ok")
            line["dedup"]["Mpairs_per_s_whole_job"] = d["pairing"].get("Mpairs_per_s_whole_job")
    line["final"] = bool(final)
    if extras is not None:
        line["extras_ok"] = all(_all_checks_hold(b) for b in extras.values())
        line["extras_failed"] = [k for k, b in extras.items() if not _all_checks_hold(b)]
        line["extras"] = extras_summary(extras)
    line = _plain(line)
    for drop in (None, "extras", "dedup", "host_fed"):
        if drop is not None:
            line.pop(drop, None)
        text = json.dumps(line, ensure_ascii=True, separators=(", ", ": "))
        if len(text) <= HEADLINE_MAX_BYTES:
            break
    assert len(text) <= HEADLINE_MAX_BYTES and "\n" not in text, len(text)
    return text


def extra_line(name, block):
    """one untimed extra as a stdout line of its own, printable ASCII"""
    return json.dumps({"extra": name, **_plain(block)}, ensure_ascii=True)



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the first passes over a fresh 35 GB image run 2 % slower than the ones behind them - 8.11 - 8.18 ms against 7.98 - 8.03
    # with five warm-up steps, whatever the number of timed steps: the defaults leave that behind)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exact", action="store_true", help="use only the wave-per-record validator")
    ap.add_argument("--two-pass", action="store_true", help="census + framing passes instead of the single-pass path")
    ap.add_argument("--no-index-extra", action="store_true",
                    help="skip the untimed extra: fastq_info default mode (validate + unique read-name index)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE.json configs[4] as written: 125 M pairs per GPU (1 B over eight), the pairing leg of the "
                         "cross-rank extra between two DISTINCT mate files (it always is; this only sets the size)")
    ap.add_argument("--no-dedup-extra", action="store_true",
                    help="skip the extra: unique read names across ALL ranks (fingerprint all-to-all over RCCL)")
    ap.add_argument("--dedup-extra", action="store_true",
                    help="run that extra at --gpus 1 too (a 1-rank RCCL group; off by default: RCCL prints a banner)")
    ap.add_argument("--no-barcodes-extra", action="store_true",
                    help="skip the extra: fastq_pre_barcodes 10x v2 layout (BASELINE.json configs[2])")
    ap.add_argument("--barcode-pairs", type=int, default=200_000_000, help="read pairs of that extra (BASELINE: 200 M)")
    ap.add_argument("--no-filters-extra", action="store_true",
                    help="skip the fastq_filter_n / fastq_trim_poly_at extra (fqg_records_filter on the same image)")
    ap.add_argument("--no-umi-extra", action="store_true",
                    help="skip the extra: bam_umi_count on BASELINE.json configs[3] (10k cells x 20k genes x 5M triples)")
    ap.add_argument("--umi-triples", type=int, default=5_000_000)
    ap.add_argument("--no-tags-extra", action="store_true", help="skip the bam_add_tags extra")
    ap.add_argument("--no-filterpair-extra", action="store_true", help="skip the fastq_filterpair extra")
    ap.add_argument("--filterpair-records", type=int, default=40_000_000, help="records per file of that extra")
    ap.add_argument("--tags-alignments", type=int, default=6_500_000)
    ap.add_argument("--no-shapes-extra", action="store_true",
                    help="skip the validate pass on mixed-length short reads and on ONT-like long reads")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the host-fed measurement (the same reads from pinned host RAM / from a tmpfs file)")
    ap.add_argument("--extras-out", default=None, metavar="PATH",
                    help="also write every untimed extra as one JSON document here (default: gpurun_out/bench_extras.json "
                         "when gpurun_out/ exists)")
    ap.add_argument("--bgzf-helper", nargs=2, metavar=("SRC", "DST"), help=argparse.SUPPRESS)
    ap.add_argument("--gz-helper", nargs=3, metavar=("SRC", "DST", "NBYTES"), help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.config4:
        a.reads = 125_000_000
    return a


def cpu_baseline(image_prefix_bytes, n_reads):
    """Time the CPU side on a bounded sample of the same workload (rank 0, N=1 only), BASELINE.md section 3:
      (A) the reference program itself (oracle/_ref/fastq_info -r), one process - the only mode the
          reference has; wall time around the process, input in the page cache, 3 runs, median;
      (B) N independent processes of it on N equal record-aligned shards (the -r path shards trivially),
          N = the host's cores (stated), every shard as long as the (A) sample, 3 runs, median.
    Falls back to the C restatement (oracle/liboracle_fq.so) when the reference binary is not there."""
    from oracle import loader as orc  # the only place bench.py touches oracle/: as the baseline

    ref = os.path.join(orc.REF_DIR, "fastq_info")
    sample = f"first {n_reads} reads of the same synthetic batch ({len(image_prefix_bytes)/1e9:.2f} GB, uncompressed)"
    if os.path.exists(ref):
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=shm) as tmp:
            path = os.path.join(tmp, "sample.fastq")
            with open(path, "wb") as f:
                f.write(image_prefix_bytes)
            with open(path, "rb") as f:  # page cache warm
                while f.read(1 << 26):
                    pass
            runs, ok = [], True
            for _ in range(3):
                t0 = time.perf_counter()
                p = subprocess.run([ref, "-r", path], capture_output=True)
                runs.append(time.perf_counter() - t0)
                ok = ok and p.returncode == 0 and (f"Number of reads: {n_reads}".encode() in p.stderr)
            dt = sorted(runs)[1]
            out = {"value": n_reads / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "reference",
                   "sample": sample + "; reference fastq_info -r, 1 process (the reference is single-threaded), "
                             "median of 3 runs",
                   "seconds": dt, "seconds_runs": runs, "ok": bool(ok)}
            # (B) all cores: N processes, each on a shard of its own (equal shards; bounded: 1/16 of the (A) sample each)
            ncpu = os.cpu_count() or 1
            if ncpu > 1:
                m = max(1000, n_reads // 16)
                rec_bytes = len(image_prefix_bytes) // n_reads
                shard = os.path.join(tmp, "shard.fastq")
                with open(shard, "wb") as f:
                    f.write(image_prefix_bytes[: m * rec_bytes])
                mruns, mok = [], True
                for _ in range(3):
                    t0 = time.perf_counter()
                    ps = [subprocess.Popen([ref, "-r", shard], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                          for _ in range(ncpu)]
                    for q in ps:
                        mok = (q.wait() == 0) and mok
                    mruns.append(time.perf_counter() - t0)
                mdt = sorted(mruns)[1]
                out["all_cores"] = {"value": m * ncpu / mdt / 1e6, "unit": "Mreads/s", "cores": ncpu, "kind": "reference",
                                    "sample": f"{ncpu} independent reference processes (one per core of the host), each "
                                              f"validating a shard of {m} reads from the page cache, median of 3 runs",
                                    "seconds": mdt, "seconds_runs": mruns, "ok": bool(mok)}
        return out
    t0 = time.perf_counter()
    r = orc.fastq_info(image_prefix_bytes, "sample.fastq", flags=orc.FLAG_R)
    dt = time.perf_counter() - t0
    return {"value": n_reads / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": sample + "; oracle/fq_oracle.c restatement", "seconds": dt, "ok": r["exit"] == 0}


def barcodes_extra(ctx, fq, torch, dev, n_pairs, programs=True):
    """fastq_pre_barcodes' main loop (fqg_barcodes_transform) on BASELINE.json configs[2]: 10x v2 layout,
    index1 = 16 bp cell barcode + 10 bp UMI, read1 = 150 bp cDNA, flags as sh/fastq2bam:125-130 builds
    them plus --min_qual 10, SAM output.  Both files are synthetic and resident in HBM; checked through
    size-independent properties (discard count recomputed with torch from the quality bytes, output
    size) and against the oracle on the first 2 000 pairs."""
    A = fq.abi
    R1, R2 = A.synth_record_bytes(26), A.synth_record_bytes(150)
    img1 = torch.empty(n_pairs * R1, dtype=torch.uint8, device=dev)
    img2 = torch.empty(n_pairs * R2, dtype=torch.uint8, device=dev)
    ctx.synth_fastq(img1.data_ptr(), n_pairs, 26, first_index=0, seed=777, mate=1)
    ctx.synth_fastq(img2.data_ptr(), n_pairs, 150, first_index=0, seed=778, mate=2)
    ctx.synchronize()
    # cell barcodes as SURVEY 8d(3) asks: drawn from a list of 10 000, one random base changed in about 1 % of the reads
    g0 = torch.Generator(device=dev)
    g0.manual_seed(4321)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    wl_codes = torch.randint(0, 4, (10_000, 16), generator=g0, device=dev)
    cb = img1.view(n_pairs, R1)[:, 45:45 + 16]
    step_rows = 25_000_000
    for a in range(0, n_pairs, step_rows):
        b = min(n_pairs, a + step_rows)
        pick = torch.randint(0, 10_000, (b - a,), generator=g0, device=dev)
        codes = wl_codes[pick]
        err = torch.nonzero(torch.rand(b - a, generator=g0, device=dev) < 0.01).squeeze(1)
        codes[err, torch.randint(0, 16, (err.numel(),), generator=g0, device=dev)] = torch.randint(0, 4, (err.numel(),), generator=g0, device=dev)
        cb[a:b] = acgt[codes]
        del pick, codes, err
    torch.cuda.synchronize()
    # barcode qualities as SURVEY 8d(3) asks: Phred 12..40, and one base below Phred 10 in about 5 % of the reads
    q = img1.view(n_pairs, R1)[:, 45 + 26 + 3: 45 + 26 + 3 + 26]
    q.clamp_(min=33 + 12)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    low = torch.rand(n_pairs, generator=g, device=dev) < 0.05
    pos = torch.randint(0, 26, (n_pairs,), generator=g, device=dev)
    rows = torch.nonzero(low).squeeze(1)
    q[rows, pos[rows]] = 33 + 5
    torch.cuda.synchronize()
    del low, pos, rows
    st1 = A.probe_first_record(bytes(img1[: 4 * R1].cpu().numpy()), True)
    st2 = A.probe_first_record(bytes(img2[: 4 * R2].cpu().numpy()), True)
    frames, states = {}, {A.READ1: st2, A.INDEX1: st1}
    t0 = time.perf_counter()
    for key, img, st, R in ((A.READ1, img2, st2, R2), (A.INDEX1, img1, st1, R1)):
        r = ctx.validate(img.data_ptr(), None, st, final=True, flags=A.VALIDATE_FRAME_ONLY, nbytes=n_pairs * R)
        assert r["n_records"] == n_pairs, r
        frames[key] = ctx.retain_frame()
    ctx.synchronize()
    frame_s = time.perf_counter() - t0

    def run():
        return ctx.barcodes_transform(frames, states, n_pairs, umi=(A.INDEX1, 16, 10), cell=(A.INDEX1, 0, 16),
                                      phred=33, min_qual=10, sam=True)

    run()
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t1 = time.perf_counter()
    r = run()
    ctx.synchronize()
    wall = time.perf_counter() - t1
    prof = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith("k_bc") and v[0] > 0}
    ctx.profile(False)
    # every base of the barcode range must reach Phred 10: recount the discards from the quality bytes
    want_disc = int((q < 33 + 10).any(dim=1).sum().item())
    kept = n_pairs - r["n_discarded"]
    ok = r["code"] == 0 and r["n_done"] == n_pairs and r["n_discarded"] == want_disc and r["n_short"] == 0
    # the first 2 000 pairs against the oracle
    from oracle import pre_barcodes_oracle as pbo

    m = min(2000, n_pairs)
    files = {"i1.fastq": bytes(img1[: m * R1].cpu().numpy()), "r1.fastq": bytes(img2[: m * R2].cpu().numpy())}
    want = pbo.run_pre_barcodes(["--read1", "r1.fastq", "--index1", "i1.fastq", "--umi_read", "index1", "--umi_offset", "16",
                                 "--umi_size", "10", "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16",
                                 "--phred_encoding", "33", "--min_qual", "10", "--sam", "--outfile1", "-"], files.get)
    body = "".join(ln + "\n" for ln in want["stdout"].splitlines() if not ln.startswith("@"))
    got = ctx.barcodes_output(0, len(body)).decode("latin-1")
    kernels_ms = sum(prof.values())
    algo_bytes = n_pairs * (R1 + R2) + r["out_bytes"][0]
    out = {
        "what": "fastq_pre_barcodes main loop (fqg_barcodes_transform), 10x v2 layout, SAM output, BASELINE.json configs[2]",
        "pairs": n_pairs, "baseline_pairs": 200_000_000, "discarded": r["n_discarded"], "sam_bytes": r["out_bytes"][0],
        "wall_ms_one_call": wall * 1e3, "Mpairs_per_s_wall": n_pairs / wall / 1e6, "kernels_ms": kernels_ms,
        "kernels_ms_breakdown": prof, "framing_both_files_s": frame_s,
        "algorithmic_GB": algo_bytes / 1e9,
        "achieved_GBps_kernels": algo_bytes / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None,
        "properties_hold": bool(ok), "expected_discarded": want_disc, "sam_bytes_per_kept_pair": r["out_bytes"][0] / max(1, kept),
        "first_2000_pairs_identical_to_oracle": got == body and want["exit"] == 0,
    }
    # the whitelist stage, on its own ("report a whitelist-membership stage separately"): the 16 cell-barcode
    # characters of every index read packed as bam_umi_count packs them and looked up in the set its --known_cells file
    # gives (valid_barcode, src/bam_umi_count.c:523-535); the count is recomputed with torch from the same bytes
    try:
        wl_bytes = acgt[wl_codes].cpu().numpy()
        wl = A.Whitelist.from_lines(ctx, b"".join(bytes(row) + b"\n" for row in wl_bytes))
        ctx.barcodes_whitelist(frames[A.INDEX1], wl, 0, 16, n_records=n_pairs)
        ctx.profile(True)
        ctx.profile_reset()
        wr = ctx.barcodes_whitelist(frames[A.INDEX1], wl, 0, 16, n_records=n_pairs)
        ctx.synchronize()
        wms = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith("k_bc_whitelist") and v[0] > 0}
        ctx.profile(False)
        pw = torch.tensor([10 ** k for k in range(16)], dtype=torch.int64, device=dev)  # (the string is packed from its end)
        lut = torch.zeros(256, dtype=torch.int64, device=dev)
        for ch, dgt in zip(b"ACGTN", range(1, 6)):
            lut[ch] = dgt
        wl_packed = ((wl_codes + 1) * pw).sum(dim=1)
        want_valid = 0
        torch.cuda.empty_cache()  # (the SAM text of 200 M pairs is still in the library's buffers: little HBM is left)
        rows_at_once = 2_000_000
        for a in range(0, n_pairs, rows_at_once):
            b = min(n_pairs, a + rows_at_once)
            want_valid += int(torch.isin((lut[cb[a:b].long()] * pw).sum(dim=1), wl_packed).sum().item())
        kms = sum(wms.values())
        out["whitelist_stage"] = {
            "what": "valid_barcode on the 16 cell-barcode characters of every index read against a 10 000-entry whitelist (fqg_barcodes_whitelist)",
            "whitelist_entries": 10_000, "reads": n_pairs, "valid": wr["n_valid"], "expected_valid": want_valid,
            "short": wr["n_short"], "ok": wr["n_valid"] == want_valid and wr["n_short"] == 0, "kernel_ms": kms,
            "Mreads_per_s": n_pairs / (kms * 1e-3) / 1e6 if kms else None,
            "algorithmic_bytes_per_read": 16 + 16,  # the barcode + the two line ends that locate it
            "achieved_GBps": 32.0 * n_pairs / (kms * 1e-3) / 1e9 if kms else None,
        }
        wl.close()
    except Exception as e:
        out["whitelist_stage"] = {"error": repr(e)[:300]}
    # FASTQ -> (cell, UMI) without the BAM round trip (fqg_barcodes_census): the packed (cell, UMI) pair of
    # every read the transform above kept - what bam_add_tags + bam_umi_count make of the tags in its name - sorted, and one
    # line per cell (reads, distinct UMIs).  Checked at full size by its invariants, on the first 2 000 pairs against the
    # composition of the three oracles (names of the oracle's SAM lines -> get_barcodes -> char2uint_64).
    try:
        kw = dict(umi=(A.INDEX1, 16, 10), cell=(A.INDEX1, 0, 16), phred=33, min_qual=10, sam=True)
        torch.cuda.empty_cache()
        z = ctx.census()
        ctx.profile(True)
        ctx.profile_reset()
        ctx.synchronize()
        tc0 = time.perf_counter()
        added = ctx.barcodes_census(z, frames, states, r["n_done"], **kw)
        n_cp, n_cc = z.finish()
        ctx.synchronize()
        cwall = time.perf_counter() - tc0
        cms = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if ("census" in k) and v[0] > 0}
        ctx.profile(False)
        lines = z.cells(n_cc)
        inv = (added == kept and n_cp == kept and int(lines[:, 1].sum()) == kept and bool((lines[:, 2] <= lines[:, 1]).all())
               and bool((lines[1:, 0] > lines[:-1, 0]).all()))
        z.close()
        # the first m pairs again, as a batch of their own, against the oracles' composition
        from oracle import bam_tags_oracle as bto
        from oracle import umi_oracle as uo
        r_small = ctx.barcodes_transform(frames, states, m, **kw)
        z2 = ctx.census()
        ctx.barcodes_census(z2, frames, states, r_small["n_done"], **kw)
        p2, c2 = z2.finish()
        ce, um = z2.pairs(p2)
        z2.close()
        # (the names as FASTQ mode writes them - tags in front, src/fastq_pre_barcodes.c:192-216: what travels into the BAM)
        want_fq = pbo.run_pre_barcodes(["--read1", "r1.fastq", "--index1", "i1.fastq", "--umi_read", "index1", "--umi_offset", "16",
                                        "--umi_size", "10", "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16",
                                        "--phred_encoding", "33", "--min_qual", "10", "--outfile1", "o.fastq.gz"], files.get)
        want_pairs = []
        for ln in want_fq["files"][1].split(b"\n"):
            if not ln.startswith(b"@STAGS_"):
                continue
            qn = ln[1:].split(b" ")[0] + b"\0"
            okb, cellb, umib, _ = bto.get_barcodes(qn, 0, len(qn))
            if okb and umib:
                want_pairs.append((uo.char2uint_64(cellb), uo.char2uint_64(umib)))
        out["census_stage"] = {
            "what": "fqg_barcodes_census + fqg_census_finish: (cell, UMI) of every kept read, sorted, one line per cell (SURVEY 8f-3)",
            "pairs": int(n_cp), "cells": int(n_cc), "kernel_ms": cms, "wall_ms": cwall * 1e3,
            "census_kernel_ms": cms.get("k_bc_census"), "Mpairs_per_s_census_kernel": (n_pairs / (cms["k_bc_census"] * 1e-3) / 1e6) if cms.get("k_bc_census") else None,
            "properties_hold": bool(inv),
            "first_2000_pairs_identical_to_oracles": sorted(zip(ce.tolist(), um.tolist())) == sorted(want_pairs) and len(want_pairs) > 0,
        }
        run()  # (the status bytes and the SAM text of the whole batch again, for what follows)
    except Exception as e:
        out["census_stage"] = {"error": repr(e)[:300]}
    for f in frames.values():
        f.release()
    # a file set without a specialised kernel (three barcode sources, as the reference's pre3 fixture has them): the
    # generic instantiations k_bc_*_tile<.., 0>
    try:
        out["generic_file_set"] = barcodes_generic(ctx, fq, torch, dev, img2, R2, min(n_pairs, 50_000_000))
    except Exception as e:
        out["generic_file_set"] = {"error": repr(e)[:300]}
    if programs:
        try:
            out["programs"] = barcodes_programs(ctx, fq, torch, img1, img2, R1, R2, n_pairs, q, kernels_ms)
        except Exception as e:
            out["programs"] = {"error": repr(e)[:300]}
    ref = os.path.join(REPO, "oracle", "_ref", "fastq_pre_barcodes")
    if os.path.exists(ref):
        ms = min(n_pairs, 1_000_000)
        with tempfile.TemporaryDirectory() as tmp:
            for name, img, R in (("i1.fastq", img1, R1), ("r1.fastq", img2, R2)):
                with open(os.path.join(tmp, name), "wb") as f:
                    f.write(bytes(img[: ms * R].cpu().numpy()))
            t2 = time.perf_counter()
            p = subprocess.run(["fastq_pre_barcodes", "--read1", "r1.fastq", "--index1", "i1.fastq", "--umi_read", "index1",
                                "--umi_offset", "16", "--umi_size", "10", "--cell_read", "index1", "--cell_offset", "0",
                                "--cell_size", "16", "--phred_encoding", "33", "--min_qual", "10", "--sam", "--outfile1", "-"],
                               executable=ref, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            secs = time.perf_counter() - t2
        out["cpu_baseline"] = {"value": ms / secs / 1e6, "unit": "Mpairs/s", "cores": 1, "kind": "reference",
                               "sample": f"first {ms} pairs of the same batch, uncompressed files, SAM to /dev/null; "
                                         "reference fastq_pre_barcodes (single-threaded)",
                               "seconds": secs, "ok": p.returncode == 0}
    return out


V2_FLAGS = ["--read1", "r1.fastq", "--index1", "i1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
            "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--min_qual", "10"]


def barcodes_generic(ctx, fq, torch, dev, img_read, R_read, n):
    """fqg_barcodes_transform with the UMI, the cell and the sample barcode in three files of their own beside the read
    (index1 / index2 / index3: the layout of the reference's pre3 fixture): no kernel is specialised for this file set"""
    import time

    from oracle import pre_barcodes_oracle as pbo

    A = fq.abi
    Ri = A.synth_record_bytes(26)
    idx = []
    for k in range(3):
        t = torch.empty(n * Ri, dtype=torch.uint8, device=dev)
        ctx.synth_fastq(t.data_ptr(), n, 26, first_index=0, seed=900 + k, mate=1)
        idx.append(t)
    ctx.synchronize()
    keys = (A.READ1, A.INDEX1, A.INDEX2, A.INDEX3)
    imgs = (img_read, idx[0], idx[1], idx[2])
    Rs = (R_read, Ri, Ri, Ri)
    frames, states = {}, {}
    for key, img, R in zip(keys, imgs, Rs):
        st = A.probe_first_record(bytes(img[: 4 * R].cpu().numpy()), True)
        r = ctx.validate(img.data_ptr(), None, st, final=True, flags=A.VALIDATE_FRAME_ONLY, nbytes=n * R)
        assert r["n_records"] == n, r
        frames[key] = ctx.retain_frame()
        states[key] = st

    def run():
        return ctx.barcodes_transform(frames, states, n, umi=(A.INDEX1, 0, 10), cell=(A.INDEX2, 0, 16), sample=(A.INDEX3, 0, 8),
                                      phred=33, min_qual=0, sam=True)

    run()
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    r = run()
    ctx.synchronize()
    wall = time.perf_counter() - t0
    prof = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith("k_bc") and v[0] > 0}
    ctx.profile(False)
    m = min(2000, n)
    names = ("r1.fastq", "i1.fastq", "i2.fastq", "i3.fastq")
    files = {nm: bytes(img[: m * R].cpu().numpy()) for nm, img, R in zip(names, imgs, Rs)}
    want = pbo.run_pre_barcodes(["--read1", "r1.fastq", "--index1", "i1.fastq", "--index2", "i2.fastq", "--index3", "i3.fastq",
                                 "--umi_read", "index1", "--umi_offset", "0", "--umi_size", "10", "--cell_read", "index2", "--cell_offset", "0",
                                 "--cell_size", "16", "--sample_read", "index3", "--sample_offset", "0", "--sample_size", "8",
                                 "--phred_encoding", "33", "--sam", "--outfile1", "-"], files.get)
    body = "".join(ln + "\n" for ln in want["stdout"].splitlines() if not ln.startswith("@"))
    got = ctx.barcodes_output(0, len(body)).decode("latin-1")
    for f in frames.values():
        f.release()
    del idx, imgs
    torch.cuda.empty_cache()  # (the programs timed next are processes of their own and need device memory too)
    kms = sum(prof.values())
    algo = n * (R_read + 3 * Ri) + r["out_bytes"][0]
    return {"what": "read1 + index1 (UMI) + index2 (cell) + index3 (sample): the generic tile kernels", "pairs": n,
            "kernels_ms": kms, "kernels_ms_breakdown": prof, "wall_ms_one_call": wall * 1e3, "algorithmic_GB": algo / 1e9,
            "achieved_GBps_kernels": algo / (kms * 1e-3) / 1e9 if kms else None,
            "ok": r["code"] == 0 and r["n_done"] == n and r["n_discarded"] == 0,
            "first_2000_pairs_identical_to_oracle": got == body and want["exit"] == 0}


def barcodes_programs(ctx, fq, torch, img1, img2, R1, R2, n_pairs, qual_rows, kernels_ms):
    """The drop-in PROGRAM on the same pairs: bin/fastq_pre_barcodes on two files in tmpfs, process start
    to exit, with both outputs the survey names - (i) --sam --outfile1 - (SAM text to stdout, here /dev/null) and
    (ii) --outfile1 out.fastq.gz (the re-tagged FASTQ through the parallel gzip of host/fq_parallel.h).  As many of
    the pairs as /dev/shm holds beside the output of (ii).  Checked: the counts the program prints against the recount
    from the quality bytes; and, on a 2 M-pair prefix, the sha256 of what (ii) inflates to against the sha256 of the
    library's FASTQ-mode output for the same pairs."""
    import gzip
    import hashlib
    import shutil

    A = fq.abi
    exe = os.path.join(REPO, "bin", "fastq_pre_barcodes")
    shm = "/dev/shm"
    if not (os.path.exists(exe) and os.path.isdir(shm)):
        return {"skipped": "no bin/fastq_pre_barcodes or no /dev/shm"}
    free = shutil.disk_usage(shm).free
    avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    per_pair_in, per_pair_gz = R1 + R2, 260  # (the gzip'd FASTQ of (ii): about 0.6 of its 400 bytes per pair)
    m = int(min(n_pairs, (min(free, avail) - (24 << 30)) // (per_pair_in + per_pair_gz)))
    if m < 1_000_000:
        return {"skipped": f"/dev/shm has {free >> 30} GiB free, the host {avail >> 30} GiB available"}
    d = tempfile.mkdtemp(prefix="fqg_bench_bc_", dir=shm)
    # (the programs are processes of their own on the same GPU: what this process's context still holds of the 200 M-pair
    # call above - 120 GB of SAM text among it - is given back first)
    ctx.release_scratch()
    torch.cuda.empty_cache()
    res = {"pairs": m, "baseline_pairs": 200_000_000, "input_GB": m * per_pair_in / 1e9,
           "host_cores_usable": len(os.sched_getaffinity(0)), "host_cores": os.cpu_count(),
           "kernels_only_ms_for_all_pairs_of_the_extra": kernels_ms}
    try:
        t0 = time.perf_counter()
        for name, img, R in (("i1.fastq", img1, R1), ("r1.fastq", img2, R2)):
            with open(os.path.join(d, name), "wb") as f:
                rows = (256 << 20) // R
                for a in range(0, m, rows):
                    f.write(img[a * R:min(m, a + rows) * R].cpu().numpy().data)
        res["write_tmpfs_files_s"] = time.perf_counter() - t0
        want_disc = int((qual_rows[:m] < 33 + 10).any(dim=1).sum().item())

        def counts(err):
            got = {}
            for ln in err.decode("latin-1").splitlines():
                for key in ("Reads processed: ", "Reads discarded: "):
                    if key in ln:
                        got[key] = int(ln.split(key)[1].split()[0])
            return got

        def timed(args, stdout, env=None):
            t = time.perf_counter()
            p = subprocess.run(["fastq_pre_barcodes"] + V2_FLAGS + args, executable=exe, cwd=d, stdout=stdout,
                               stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
            return time.perf_counter() - t, p

        legs = {}
        m_all = m
        for label, args, sink in (("sam_to_stdout", ["--sam", "--outfile1", "-"], subprocess.DEVNULL),
                                  ("fastq_gz_file", ["--outfile1", "out.fastq.gz"], None)):
            if sink is None and m_all > 50_000_000:
                # gzip'ing the re-tagged FASTQ is host work, minutes of it at 200 M pairs on a few cores: this leg takes the
                # first 50 M pairs (the files are written again, shorter) and runs once per deflate level
                m = 50_000_000
                for name, img, R in (("i1.fastq", img1, R1), ("r1.fastq", img2, R2)):
                    with open(os.path.join(d, name), "wb") as f:
                        rows = (256 << 20) // R
                        for a in range(0, m, rows):
                            f.write(img[a * R:min(m, a + rows) * R].cpu().numpy().data)
                want_disc = int((qual_rows[:m] < 33 + 10).any(dim=1).sum().item())
            runs = []
            for _ in range(2 if sink is not None else 1):
                secs, p = timed(args, sink, {"FQGPU_TIMING": "1"})
                c = counts(p.stderr)
                says = [ln[ln.find("fqgpu timing"):] for ln in p.stderr.decode("latin-1").splitlines() if "fqgpu timing" in ln]
                ok = p.returncode == 0 and c.get("Reads processed: ") == m and c.get("Reads discarded: ") == want_disc
                runs.append(secs)
                if not ok:
                    break
            best = min(runs)
            if not ok:  # (what the program said: a leg that fails must say why)
                says = says + ["exit status %d" % p.returncode, p.stderr.decode("latin-1")[-600:]]
            legs[label] = {"pairs": m, "seconds": runs, "Mpairs_per_s": m / best / 1e6, "input_GBps": m * per_pair_in / best / 1e9, "ok": ok, "says": says,
                           "includes": "process start, HIP initialisation, pinned slots, reading both files, H2D, kernels, D2H, "
                                       + ("parallel gzip (level as the reference's gzopen \"w\"), file written to tmpfs" if sink is None else "SAM text written to /dev/null")}
            if sink is None and os.path.exists(os.path.join(d, "out.fastq.gz")):
                legs[label]["output_gz_GB"] = os.path.getsize(os.path.join(d, "out.fastq.gz")) / 1e9
                os.unlink(os.path.join(d, "out.fastq.gz"))
                # deflate is the host's whole cost here: the same run with the members from host/fq_fastdeflate.h instead of
                # zlib's (FQGPU_GZIP_FAST=1: zlib level 1's size class; what a reader inflates is the same - the prefix check
                # below runs with it too)
                secs, p = timed(args, sink, {"FQGPU_GZIP_FAST": "1", "FQGPU_TIMING": "1"})
                c = counts(p.stderr)
                legs[label]["with_FQGPU_GZIP_FAST"] = {
                    "seconds": secs, "Mpairs_per_s": m / secs / 1e6,
                    "ok": p.returncode == 0 and c.get("Reads processed: ") == m and c.get("Reads discarded: ") == want_disc,
                    "output_gz_GB": os.path.getsize(os.path.join(d, "out.fastq.gz")) / 1e9 if os.path.exists(os.path.join(d, "out.fastq.gz")) else None,
                    "says": [ln[ln.find("fqgpu timing"):] for ln in p.stderr.decode("latin-1").splitlines() if "fqgpu timing" in ln][:1]}
                if os.path.exists(os.path.join(d, "out.fastq.gz")):
                    os.unlink(os.path.join(d, "out.fastq.gz"))
        res["legs"] = legs
        # the bytes of (ii) on a prefix: program -> gunzip -> sha256 against the library's own FASTQ-mode output
        k = min(m, 2_000_000)
        for name, img, R in (("i1.fastq", img1, R1), ("r1.fastq", img2, R2)):
            with open(os.path.join(d, name), "wb") as f:
                f.write(img[: k * R].cpu().numpy().data)
        secs, p = timed(["--outfile1", "out.fastq.gz"], None)
        with gzip.open(os.path.join(d, "out.fastq.gz"), "rb") as f:
            prog_sha = hashlib.sha256(f.read()).hexdigest()
        secs, p_fast = timed(["--outfile1", "out.fastq.gz"], None, {"FQGPU_GZIP_FAST": "1"})
        with gzip.open(os.path.join(d, "out.fastq.gz"), "rb") as f:
            fast_sha = hashlib.sha256(f.read()).hexdigest()
        st1 = A.probe_first_record(bytes(img1[: 4 * R1].cpu().numpy()), True)
        st2 = A.probe_first_record(bytes(img2[: 4 * R2].cpu().numpy()), True)
        frames = {}
        for key, img, st, R in ((A.READ1, img2, st2, R2), (A.INDEX1, img1, st1, R1)):
            r = ctx.validate(img.data_ptr(), None, st, final=True, flags=A.VALIDATE_FRAME_ONLY, nbytes=k * R)
            assert r["n_records"] == k, r
            frames[key] = ctx.retain_frame()
        r = ctx.barcodes_transform(frames, {A.READ1: st2, A.INDEX1: st1}, k, umi=(A.INDEX1, 16, 10), cell=(A.INDEX1, 0, 16),
                                   phred=33, min_qual=10, sam=False)
        lib_sha = hashlib.sha256(ctx.barcodes_output(1, r["out_bytes"][1])).hexdigest()
        for f in frames.values():
            f.release()
        res["prefix_check"] = {"pairs": k, "program_output_sha256": prog_sha, "library_output_sha256": lib_sha,
                               "with_FQGPU_GZIP_FAST_sha256": fast_sha,
                               "identical": p.returncode == 0 and p_fast.returncode == 0 and prog_sha == lib_sha == fast_sha}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


def filters_extra(ctx, fq, torch, dev, image, n, R, st, read_len):
    """fastq_filter_n and fastq_trim_poly_at record loops (fqg_records_filter) on the bench's own
    image.  Checked at full size through counts recomputed with torch from the sequence bytes and output
    sizes, and on the first 2 000 records against the oracle; the reference programs are timed on a sample."""
    A = fq.abi
    r = ctx.validate(image.data_ptr(), None, st, final=True, flags=A.VALIDATE_FRAME_ONLY, nbytes=n * R)
    assert r["n_records"] == n, r
    frame = ctx.retain_frame()
    hdr = R - 2 * read_len - 4  # '@name\n' ; then seq\n+\n qual\n
    seq = image.view(n, R)[:, hdr: hdr + read_len]
    is_n = (seq == ord("N"))
    out = {"what": "fastq_filter_n / fastq_trim_poly_at record loops (fqg_records_filter) on the same image", "reads": n}
    from oracle import filter_oracle as fo

    m = min(2000, n)
    head = bytes(image[: m * R].cpu().numpy())
    modes = (("filter_n", dict(mode=A.FILTER_N, max_n_percent=0), ["in.fastq"], fo.filter_n, "stdout"),
             ("trim_poly_at", dict(mode=A.FILTER_POLY_AT, min_poly_at_len=3, min_len=10),
              ["--file", "in.fastq", "--outfile", "x", "--min_poly_at_len", "3"], fo.trim_poly_at, "out"))
    for name, kw, argv, oracle, field in modes:
        ctx.records_filter(frame, n, **kw)
        ctx.profile(True)
        ctx.profile_reset()
        ctx.synchronize()
        t1 = time.perf_counter()
        fr = ctx.records_filter(frame, n, **kw)
        ctx.synchronize()
        wall = time.perf_counter() - t1
        prof = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith("k_rf") and v[0] > 0}
        ctx.profile(False)
        if name == "filter_n":
            want_disc = int(is_n.any(dim=1).sum().item())
            ok = fr["n_discarded"] == want_disc and fr["out_bytes"] == (n - want_disc) * R and fr["n_trimmed"] == 0
        else:
            a_or_n = (seq[:, -3:] == ord("A")) | is_n[:, -3:]
            t_or_n = (seq[:, :3] == ord("T")) | is_n[:, :3]
            want_trim = int((a_or_n.all(dim=1) | t_or_n.all(dim=1)).sum().item())
            ok = fr["n_trimmed"] == want_trim and fr["n_discarded"] == 0 and fr["out_bytes"] < n * R
        want = oracle(argv, lambda p: head)
        got = ctx.records_filter_output(len(want[field]))
        kernels_ms = sum(prof.values())
        algo = n * R + fr["out_bytes"]
        out[name] = {"wall_ms_one_call": wall * 1e3, "Mreads_per_s_wall": n / wall / 1e6, "kernels_ms": kernels_ms,
                     "kernels_ms_breakdown": prof, "result": fr, "algorithmic_GB": algo / 1e9,
                     "achieved_GBps_kernels": algo / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None,
                     "properties_hold": bool(ok), "first_2000_records_identical_to_oracle": got == want[field]}
    frame.release()
    with tempfile.TemporaryDirectory() as tmp:
        for name, ms, argv in (("fastq_filter_n", min(n, 4_000_000), ["in.fastq"]),
                               ("fastq_trim_poly_at", min(n, 1_000_000),
                                ["--file", "in.fastq", "--outfile", "o.fastq.gz", "--min_poly_at_len", "3"])):
            ref = os.path.join(REPO, "oracle", "_ref", name)
            if not os.path.exists(ref):
                continue
            with open(os.path.join(tmp, "in.fastq"), "wb") as f:
                f.write(bytes(image[: ms * R].cpu().numpy()))
            t2 = time.perf_counter()
            p = subprocess.run([name] + argv, executable=ref, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            secs = time.perf_counter() - t2
            out[name.replace("fastq_", "")]["cpu_baseline"] = {
                "value": ms / secs / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "reference",
                "sample": f"first {ms} reads of the same image, uncompressed input; reference {name} (single-threaded"
                          + (", gzip level 4 output included)" if "trim" in name else ", output to /dev/null)"),
                "seconds": secs, "ok": p.returncode == 0}
    return out


def e2e_block(ctx, fq, torch, dev, image, n, R, st):
    """BASELINE.json configs[1] says "uncompressed in host RAM": the same reads fed from the host.
      abi: the image in pinned host memory, handed to fqg_validate(FQG_MEM_HOST) in 1 GiB pieces (H2D over
           PCIe + kernels per piece; nothing to read)
      cli: bin/fastq_info -r on a file in tmpfs - the whole drop-in program: process start, HIP
           initialisation, a ring of pinned slots filled by pread() threads ahead of the GPU
           (fastq_utils_amd/host/fq_input.h), H2D, kernels, summary
    The headline `value` stays the HBM-resident rate; these are the end-to-end rates next to it."""
    import shutil

    nbytes = n * R
    avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    if avail < 3 * nbytes:
        return {"skipped": f"host has {avail >> 30} GiB available, the measurement wants 3 x {nbytes >> 30} GiB"}
    t0 = time.perf_counter()
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    host.copy_(image[:nbytes])
    torch.cuda.synchronize()
    out = {"reads": n, "bytes": nbytes, "pin_and_copy_to_host_s": time.perf_counter() - t0,
           "host_cores_usable": len(os.sched_getaffinity(0)), "host_cores": os.cpu_count()}
    piece = (1 << 30) // R * R
    best = None
    for _ in range(2):
        acc = ctx.accumulator()
        t0 = time.perf_counter()
        done = 0
        for off in range(0, nbytes, piece):
            nb = min(piece, nbytes - off)
            r = ctx.validate(host.data_ptr() + off, acc, st, final=(off + nb == nbytes), nbytes=nb, mem=fq.abi.MEM_HOST)
            assert r["code"] == 0, r
            done += r["n_records"]
        ctx.synchronize()
        dt = time.perf_counter() - t0
        assert done == n and acc.read()["num_rds"] == n
        acc.close()
        best = dt if best is None else min(best, dt)
    out["abi_pinned_host_image"] = {"seconds": best, "Mreads_per_s": n / best / 1e6, "PCIe_GBps": nbytes / best / 1e9,
                                    "piece_bytes": piece, "runs": 2}
    exe = os.path.join(REPO, "bin", "fastq_info")
    shm = "/dev/shm"
    if os.path.exists(exe) and os.path.isdir(shm) and shutil.disk_usage(shm).free > nbytes + (1 << 30):
        path = os.path.join(shm, "fqg_bench_%d.fastq" % os.getpid())
        try:
            t0 = time.perf_counter()
            arr = host.numpy()
            with open(path, "wb") as f:
                for off in range(0, nbytes, 1 << 30):
                    f.write(arr[off:off + (1 << 30)].data)
            out["write_tmpfs_file_s"] = time.perf_counter() - t0
            runs = []
            for _ in range(3):
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-r", path], capture_output=True)
                runs.append(time.perf_counter() - t0)
                assert p.returncode == 0, p.stderr[-300:]
                assert ("Number of reads: %d" % n).encode() in p.stderr, p.stderr[-300:]
            med = sorted(runs)[1]
            threads = min(12, os.cpu_count() or 1)
            # where the time goes (FQGPU_TIMING=1 makes the program say), and two other stager shapes
            variants = {}
            for label, env in (("timing", {"FQGPU_TIMING": "1"}), ("32_threads", {"FQGPU_HOST_THREADS": "32", "FQGPU_TIMING": "1"}),
                               ("256MiB_slots", {"FQGPU_CHUNK_MB": "256", "FQGPU_TIMING": "1"}),
                               ("two_contexts_one_gpu", {"FQGPU_DEVICES": "0,0"}),
                               # (what the program does by itself on a file of this size since round 6; and the loop of one
                               # context it ran before)
                               ("one_context", {"FQGPU_ONE_CONTEXT": "1", "FQGPU_TIMING": "1"})):
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-r", path], capture_output=True, env=dict(os.environ, **env))
                dt = time.perf_counter() - t0
                err = p.stderr.decode("latin-1")
                variants[label] = {"seconds": dt, "Mreads_per_s": n / dt / 1e6, "ok": p.returncode == 0,
                                   "says": [ln[ln.find("fqgpu timing"):] for ln in err.splitlines() if "fqgpu timing" in ln]}
            out["negative_controls"] = negative_controls(exe, path, arr, n, R)
            # the inputs people have are compressed: a bgzip'd copy of the same file (BGZF blocks, inflated on all cores
            # by host/fq_input.h) through the same program
            try:
                bgz = path + ".gz"
                t0 = time.perf_counter()
                # (in a process of its own: a pool of workers must not be forked from a process that holds the GPU)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--bgzf-helper", path, bgz], check=True)
                bgz_bytes = os.path.getsize(bgz)
                made = time.perf_counter() - t0
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-r", bgz], capture_output=True, env=dict(os.environ, FQGPU_TIMING="1"))
                dt = time.perf_counter() - t0
                out["cli_fastq_info_r_bgzf_file"] = {
                    "seconds": dt, "Mreads_per_s": n / dt / 1e6, "inflated_GBps": nbytes / dt / 1e9, "compressed_GB": bgz_bytes / 1e9,
                    "ok": p.returncode == 0 and ("Number of reads: %d" % n).encode() in p.stderr,
                    "times_the_plain_file": dt / med, "made_in_s": made,
                    "what": "the same reads as a bgzip'd file (64 KiB BGZF blocks, zlib level 1) in tmpfs; blocks inflated on every "
                            "core while the GPU validates the previous piece; a single-member .gz stays on one zlib thread",
                    "says": [ln[ln.find("fqgpu timing"):] for ln in p.stderr.decode("latin-1").splitlines() if "fqgpu timing" in ln]}
            except Exception as e:
                out["cli_fastq_info_r_bgzf_file"] = {"error": repr(e)[:300]}
            finally:
                if os.path.exists(path + ".gz"):
                    os.unlink(path + ".gz")
            # ... and what most people have is ONE gzip member (gzip, pigz, bcl2fastq): the first 10 M reads as such a file
            # (written the way pigz writes: independent deflate runs between flush points), read by the chunked many-core
            # reader (host/fq_pgzip.h) and, for comparison, by one zlib thread as the reference reads it
            try:
                gz = path + ".1member.gz"
                n_gz = min(n, int(os.environ.get("FQGPU_BENCH_GZ_READS", "10000000")))  # (all 100 M: a member beyond 4 GiB, 45 s to make)
                t0 = time.perf_counter()
                subprocess.run([sys.executable, os.path.abspath(__file__), "--gz-helper", path, gz, str(n_gz * R)], check=True)
                made = time.perf_counter() - t0
                leg = {"reads": n_gz, "inflated_GB": n_gz * R / 1e9, "compressed_GB": os.path.getsize(gz) / 1e9, "made_in_s": made,
                       "what": "the first reads of the same file as a single-member .gz (deflate level 1, flush points every 32 MiB "
                               "as pigz makes them) in tmpfs, through bin/fastq_info -r: chunks of the compressed bytes inflated side "
                               "by side (block starts searched for, 32 KiB windows resolved afterwards), then by one zlib thread"}
                for label, env in (("by_chunks", {}), ("one_zlib_thread", {"FQGPU_NO_PARALLEL_INFLATE": "1"})):
                    t0 = time.perf_counter()
                    p = subprocess.run([exe, "-r", gz], capture_output=True, env=dict(os.environ, FQGPU_TIMING="1", **env))
                    dt = time.perf_counter() - t0
                    leg[label] = {"seconds": dt, "Mreads_per_s": n_gz / dt / 1e6, "inflated_GBps": n_gz * R / dt / 1e9,
                                  "ok": p.returncode == 0 and ("Number of reads: %d" % n_gz).encode() in p.stderr,
                                  "says": [ln[ln.find("fqgpu timing"):] for ln in p.stderr.decode("latin-1").splitlines()
                                           if "inflated by chunks" in ln or "pieces" in ln]}
                leg["speedup"] = leg["one_zlib_thread"]["seconds"] / leg["by_chunks"]["seconds"]
                out["cli_fastq_info_r_gz_file"] = leg
            except Exception as e:
                out["cli_fastq_info_r_gz_file"] = {"error": repr(e)[:300]}
            finally:
                if os.path.exists(path + ".1member.gz"):
                    os.unlink(path + ".1member.gz")
            out["cli_fastq_info_r_tmpfs_file"] = {
                "seconds_median_of_3": med, "seconds": runs, "Mreads_per_s": n / med / 1e6, "GBps": nbytes / med / 1e9,
                "stager": f"3 pinned slots of 128 MiB, a pool of {threads} pread threads (FQGPU_CHUNK_MB / FQGPU_HOST_THREADS)",
                "includes": "process start, HIP initialisation, pinned allocation, read, H2D, kernels, summary",
                "variants": variants}
        finally:
            if os.path.exists(path):
                os.unlink(path)
    else:
        out["cli_fastq_info_r_tmpfs_file"] = {"skipped": "no bin/fastq_info or not enough room in /dev/shm"}
    return out


def _crc32_combine(crc1, crc2, len2):
    """CRC-32 of A + B from crc32(A), crc32(B) and len(B): crc1 pushed through len2 zero bytes - the operator for one
    zero bit, squared as often as len2 has binary digits - then xor crc2 (what zlib's crc32_combine computes)"""
    def times(mat, vec):
        acc, i = 0, 0
        while vec:
            if vec & 1:
                acc ^= mat[i]
            vec >>= 1
            i += 1
        return acc

    if len2 <= 0:
        return crc1
    op = [0xEDB88320] + [1 << k for k in range(31)]  # one zero bit
    for _ in range(3):
        op = [times(op, op[k]) for k in range(32)]  # ... two, four, eight bits: one zero byte
    while len2:
        if len2 & 1:
            crc1 = times(op, crc1)
        len2 >>= 1
        if len2:
            op = [times(op, op[k]) for k in range(32)]
    return crc1 ^ crc2


def _gz_slice(args):
    import zlib

    path, a, b, last = args
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    blob = c.compress(data) + (c.flush() if last else c.flush(zlib.Z_SYNC_FLUSH))
    return blob, zlib.crc32(data) & 0xFFFFFFFF, len(data)


def gz_compress_file(src, dst, nbytes):
    """the first nbytes of src as ONE gzip member, deflated by a pool of processes the way pigz does it: every slice an
    independent run of deflate blocks that ends in a flush point, the slices back to back, one trailer"""
    import struct
    from concurrent.futures import ProcessPoolExecutor

    step = 32 << 20
    tasks = [(src, a, min(nbytes, a + step), a + step >= nbytes) for a in range(0, nbytes, step)]
    crc, total = 0, 0
    with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex, open(dst, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for blob, c, ln in ex.map(_gz_slice, tasks, chunksize=1):
            f.write(blob)
            crc = _crc32_combine(crc, c, ln) if total else c
            total += ln
        f.write(struct.pack("<II", crc, total & 0xFFFFFFFF))


def _bgzf_slice(args):
    import struct
    import zlib

    path, a, b = args
    out = []
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    for i in range(0, len(data), 0xFF00):
        chunk = data[i:i + 0xFF00]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def bgzf_compress_file(src, dst):
    """src as a BGZF file (what `bgzip` writes: SAM/BAM specification 4.1), compressed by a pool of processes"""
    from concurrent.futures import ProcessPoolExecutor

    size = os.path.getsize(src)
    step = 0xFF00 * 1024  # 64 MiB of input per task
    tasks = [(src, a, min(size, a + step)) for a in range(0, size, step)]
    total = 0
    with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex, open(dst, "wb") as f:
        for blob in ex.map(_bgzf_slice, tasks, chunksize=1):
            f.write(blob)
            total += len(blob)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return total + 28


def shapes_extra(ctx, fq, torch, dev, target_bytes=8 << 30):
    """The validate pass on inputs that are less friendly to the single-pass kernels than fixed 150 bp reads:
    (a) short reads of mixed length (30-150 bp), (b) ONT-like reads of 2-20 kb, where most 4 KiB chunks hold no
    "+" line, so the line type cannot be speculated and the chunks are re-checked once their rank is known
    (k_stream_redo).  A seeded 64 MiB block generated on the host is repeated in HBM up to `target_bytes` (names
    repeat: this is the -r pass, which does not look at them)."""
    import numpy as np

    out = {}
    for tag, lo, hi in (("short_reads_30_150bp", 30, 150), ("long_reads_2_20kb", 2000, 20000)):
        rng = np.random.default_rng(99)
        parts, size, i = [], 0, 0
        while size < (64 << 20):
            L = int(rng.integers(lo, hi + 1))
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes()
            qual = (rng.integers(2, 41, L) + 33).astype(np.uint8).tobytes()
            r = b"@SHP:1:FC:1:%d:%d:%d 1:N:0:ACGT\n" % (i % 97, L, i) + seq + b"\n+\n" + qual + b"\n"
            parts.append(r)
            size += len(r)
            i += 1
        block = b"".join(parts)
        reps = max(1, target_bytes // len(block))
        tb = torch.frombuffer(bytearray(block), dtype=torch.uint8).to(dev)
        image = tb.repeat(reps)
        del tb
        n_reads = i * reps
        st = fq.abi.probe_first_record(block[:100000], True)
        acc = ctx.accumulator()
        ctx.validate(image.data_ptr(), acc, st, final=True, nbytes=image.numel())  # warm-up
        acc.reset()
        ctx.profile(True)
        ctx.profile_reset()
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = 3
        for _ in range(steps):
            r = ctx.validate(image.data_ptr(), acc, st, final=True, nbytes=image.numel())
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = {k: v[1] / steps for k, v in ctx.profile_read().items() if k.startswith("k_") and v[0] > 0}
        ctx.profile(False)
        s_ = acc.read()
        ok = r["code"] == 0 and r["n_records"] == n_reads and s_["num_rds"] == n_reads * steps and \
            s_["min_rl"] >= lo + 1 and s_["max_rl"] <= hi + 1
        kms = sum(prof.values())
        out[tag] = {"reads": n_reads, "bytes": int(image.numel()), "ms_per_pass": dt * 1e3, "Mreads_per_s": n_reads / dt / 1e6,
                    "GBps_wall": image.numel() / dt / 1e9, "kernels_ms_per_pass": prof, "GBps_kernels_only": image.numel() / (kms * 1e-3) / 1e9,
                    "path": r["path"], "ok": bool(ok)}
        acc.close()
        del image
        torch.cuda.empty_cache()
    return out


def negative_controls(exe, path, arr, n, R):
    """SURVEY 8(d)-2 at full size: one bad base / one duplicated name injected into the tmpfs file (a few bytes
    written in place, restored afterwards); the drop-in program must stop with the error line the oracle prints for
    the same defect (the oracle reads a 3-record window around it; only the line number is shifted)."""
    from oracle import loader as orc  # the checker, not the thing measured

    res = {}
    k = n - 12345 if n > 20000 else n // 2          # deep into the file
    rec = lambda i: bytes(arr[i * R:(i + 1) * R])

    def oracle_line(window, first_record, flags):
        o = orc.fastq_info(window, path, flags=flags)
        line = [ln for ln in o["stderr"].splitlines() if ln.startswith("ERROR:")]
        if not line:
            return None
        import re
        return re.sub(r"line (\d+)", lambda m: "line %d" % (int(m.group(1)) + 4 * first_record), line[0])

    def patched(offset, data, args, flags, window_first, window):
        with open(path, "r+b") as f:
            f.seek(offset)
            old = f.read(len(data))
            f.seek(offset)
            f.write(data)
        try:
            p = subprocess.run([exe] + args + [path], capture_output=True)
        finally:
            with open(path, "r+b") as f:
                f.seek(offset)
                f.write(old)
        got = [ln for ln in p.stderr.decode("latin-1").splitlines() if ln.startswith("ERROR:")]
        want = oracle_line(window, window_first, flags)
        return {"exit": p.returncode, "line": got[0] if got else None, "oracle_line": want,
                "same_line_as_oracle": bool(got) and got[0] == want and p.returncode == 3}

    # (a) a base outside the alphabet in the middle of record k's sequence
    r0 = rec(k)
    seq_at = r0.index(b"\n") + 1 + 70
    bad = bytearray(rec(k - 1) + r0 + rec(k + 1))
    bad[R + seq_at] = ord("X")
    res["bad_base"] = patched(k * R + seq_at, b"X", ["-r"], orc.FLAG_R, k - 1, bytes(bad))
    res["bad_base"]["record"] = k
    # (b) record k gets the name of record j: the unique-name index must report k
    j = k - 777
    hdr = rec(j)[: rec(j).index(b"\n")]
    if len(hdr) == r0.index(b"\n"):
        win = bytearray(rec(j) + rec(k - 1) + r0)
        win[2 * R:2 * R + len(hdr)] = hdr
        d = patched(k * R, hdr, [], 0, 0, bytes(win))
        # the window holds records j, k-1, k as records 0, 1, 2: its line 12 is line 4 (k + 1) of the file
        if d["oracle_line"]:
            d["oracle_line"] = d["oracle_line"].replace("line 12:", "line %d:" % (4 * (k + 1)))
            d["same_line_as_oracle"] = d["line"] == d["oracle_line"] and d["exit"] == 3
        d["records"] = [j, k]
        res["duplicate_name"] = d
    return res


def filterpair_extra(ctx, fq, torch, dev, n, read_len):
    """fastq_filterpair's pairing + partition (src/fastq_filterpair.c:38-228) through the bulk calls the
    drop-in program makes: file 1's names into the index, file 2's records probe and take them
    (fqg_index_probe_delete), what is left of file 1 (fqg_index_alive), and the ordered gather of the three outputs
    (fqg_records_gather).  Two synthetic files of `n` records, 90 % of them mates of each other (file 2 starts with the
    reads file 1 lacks, file 1 ends with the reads file 2 lacks), HBM-resident; the same calls on files of 3 000
    records are compared with the oracle byte for byte."""
    import numpy as np

    A = fq.abi
    R = A.synth_record_bytes(read_len)
    shift = n // 10

    def run(m, sh, want_text):
        img = [torch.empty(m * R + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        ctx.synth_fastq(img[0].data_ptr(), m, read_len, sh, 4242, 1)  # file 1: reads sh .. m + sh - 1
        ctx.synth_fastq(img[1].data_ptr(), m, read_len, 0, 4242, 2)   # file 2: reads 0 .. m - 1
        torch.cuda.synchronize()
        st = [A.probe_first_record(bytes(img[k][:4096].cpu().numpy()), True) for k in range(2)]
        ctx.profile(True)
        ctx.profile_reset()
        ctx.synchronize()
        t0 = time.perf_counter()
        # (framed as the program frames them, host/fastq_filterpair.cpp: FQG_VALIDATE_NAMES - the header lines come to the
        # index calls as capture records of the streaming pass)
        v1 = ctx.validate(img[0].data_ptr(), None, st[0], final=True, flags=A.VALIDATE_NO_STATS | A.VALIDATE_NAMES, nbytes=m * R)
        idx = ctx.name_index(m)
        ir = idx.insert_unique(st[0])
        v2 = ctx.validate(img[1].data_ptr(), None, st[1], final=True, flags=A.VALIDATE_NO_STATS | A.VALIDATE_NAMES, nbytes=m * R)
        pr, match = idx.probe_delete_np(st[1], m)
        f2 = ctx.retain_frame()
        ctx.frame_make_current(f2)
        alive = idx.alive_np(m)
        found = match < np.uint64(0xFFFFFFFFFFFFFFFE)
        # file 1's leftovers are read on from behind the LAST record that was copied (src/fastq_filterpair.c:196-216)
        k_found = np.nonzero(found)[0]
        next1 = int(match[k_found[-1]]) + 1 if k_found.size else 0
        left = np.nonzero(alive)[0]
        lists = [(idx.frame(0), match[found]), (f2, k_found), (f2, np.nonzero(~found)[0]), (idx.frame(0), left[left >= next1])]
        outs = [ctx.records_gather(fr, rec, want_output=want_text) for fr, rec in lists]
        ctx.synchronize()
        wall = time.perf_counter() - t0
        prof = {k: v[1] for k, v in ctx.profile_read().items() if v[0] > 0 and k.startswith(("k_index", "k_names", "k_gather", "k_scan64"))}
        ctx.profile(False)
        res = {"codes": [v1["code"], ir["code"], v2["code"], pr["code"]], "matched": int(found.sum()), "alive": int(alive.sum()),
               "bytes": [o[0] for o in outs], "wall": wall, "prof": prof, "text": [o[1] for o in outs],
               "images": img if want_text else None}
        f2.release()
        idx.close()
        return res

    small = run(3000, 300, True)
    from oracle import loader as orc
    b1, b2 = (bytes(small["images"][k][: 3000 * R].cpu().numpy()) for k in range(2))
    want = orc.fastq_filterpair(b1, "a_1.fastq", b2, "a_2.fastq", False)
    same = (want["exit"] == 0 and small["text"][0] == want["files"][0] and small["text"][1] == want["files"][1]
            and small["text"][2] + small["text"][3] == want["files"][2])
    run(n, shift, False)  # warm-up (allocations)
    big = run(n, shift, False)
    kernels_ms = sum(big["prof"].values())
    gathered = sum(big["bytes"])
    # names in (two header lines of ~45 bytes per pair of records), one 8-byte slot written and read, records in and out
    alg = 2 * n * 45 + 2 * n * 8 + 2 * gathered
    # the program: two tmpfs files of 10 M records, three .fastq.gz outputs (gzip members on every core; the reference
    # - and this program until round 4 - writes them through one gzwrite thread), against the reference on 500 000
    program = None
    exe = os.path.join(REPO, "bin", "fastq_filterpair")
    ref = os.path.join(REPO, "oracle", "_ref", "fastq_filterpair")
    shm = "/dev/shm"
    m = min(n, 10_000_000)
    if os.path.exists(exe) and os.path.isdir(shm) and shutil.disk_usage(shm).free > 4 * m * R + (8 << 30):
        d = tempfile.mkdtemp(prefix="fqg_fp_", dir=shm)
        try:
            img = [torch.empty(m * R + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
            ctx.synth_fastq(img[0].data_ptr(), m, read_len, m // 10, 4242, 1)
            ctx.synth_fastq(img[1].data_ptr(), m, read_len, 0, 4242, 2)
            torch.cuda.synchronize()
            for k, name in enumerate(("a_1.fastq", "a_2.fastq")):
                with open(os.path.join(d, name), "wb") as f:
                    f.write(img[k][: m * R].cpu().numpy().data)
            args = ["a_1.fastq", "a_2.fastq", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"]
            t0 = time.perf_counter()
            p = subprocess.run(["fastq_filterpair"] + args, executable=exe, cwd=d, capture_output=True)
            secs = time.perf_counter() - t0
            sizes = [os.path.getsize(os.path.join(d, o)) for o in args[2:] if os.path.exists(os.path.join(d, o))]
            program = {"records_per_file": m, "seconds": secs, "Mrecords_per_s": 2 * m / secs / 1e6, "ok": p.returncode == 0,
                       "output_gz_GB": sum(sizes) / 1e9}
            if os.path.exists(ref):
                k = 500_000
                small_img = [torch.empty(k * R + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
                ctx.synth_fastq(small_img[0].data_ptr(), k, read_len, k // 10, 4242, 1)  # (the same shape: nine in ten are mates)
                ctx.synth_fastq(small_img[1].data_ptr(), k, read_len, 0, 4242, 2)
                torch.cuda.synchronize()
                for name, i in (("b_1.fastq", 0), ("b_2.fastq", 1)):
                    with open(os.path.join(d, name), "wb") as f:
                        f.write(small_img[i][: k * R].cpu().numpy().data)
                t0 = time.perf_counter()
                pr = subprocess.run(["fastq_filterpair", "b_1.fastq", "b_2.fastq", "q1.fastq.gz", "q2.fastq.gz", "uq.fastq.gz"], executable=ref,
                                    cwd=d, capture_output=True)
                rs = time.perf_counter() - t0
                program["reference_on_500k_records_per_file"] = {"seconds": rs, "Mrecords_per_s": 2 * k / rs / 1e6, "ok": pr.returncode == 0}
            del img
            torch.cuda.empty_cache()
        except Exception as e:
            program = {"error": repr(e)[:300]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"what": "fastq_filterpair: index file 1, probe + take with file 2, ordered gather of paired / unpaired records",
            "program": program,
            "records_per_file": n, "mates": n - shift, "matched": big["matched"], "left_in_file1": big["alive"],
            "as_expected": big["codes"] == [0, 0, 0, 13] and  # (13: FQG_E_UNPAIRED - file 2 has reads without a mate)
                           big["matched"] == n - shift and big["alive"] == shift
                           and big["bytes"] == [(n - shift) * R, (n - shift) * R, shift * R, shift * R],
            "codes": big["codes"], "gathered_bytes": big["bytes"],
            "first_3000_records_identical_to_oracle": bool(same),
            "wall_ms_all_calls_incl_host_lists": big["wall"] * 1e3, "kernels_ms": kernels_ms, "kernels_ms_breakdown": big["prof"],
            "Mpairs_per_s_kernels_only": n / (kernels_ms * 1e-3) / 1e6 if kernels_ms else None,
            "algorithmic_GB": alg / 1e9, "achieved_GBps_kernels": alg / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None}


def bam_tags_extra(ctx, torch, dev, n):
    """bam_add_tags' alignment loop (fqg_bam_add_tags) on `n` synthetic alignments whose names carry a
    16-base cell and a 10-base UMI the way fastq_pre_barcodes writes them, --tx over 300 references, records resident
    in HBM.  Checked against the oracle on the first 2 000 alignments and by size at full length."""
    import numpy as np

    from tests import bamgen  # generator only

    rng = np.random.default_rng(77)
    refs = tuple((b"ENST%011d" % i, 1000) for i in range(300))
    tid = rng.integers(-1, len(refs), n).astype(np.int32)
    rec = bamgen.tagged_name_records(rng.integers(0, 1 << 32, n).astype(np.uint64), rng.integers(0, 1 << 20, n).astype(np.uint64), tid)
    hdr = bamgen.header(refs)
    R = bamgen.TAGGED_NAME_REC_BYTES
    stream = torch.empty(len(hdr) + rec.size + 64, dtype=torch.uint8, device=dev)
    stream[: len(hdr)] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev)
    stream[len(hdr): len(hdr) + rec.size] = torch.from_numpy(rec.reshape(-1)).to(dev)
    torch.cuda.synchronize()
    offs = _OffsetArray(np.arange(n, dtype=np.uint64) * np.uint64(R) + np.uint64(len(hdr)))
    names = [r[0] for r in refs]

    def run(want_output):
        return ctx.bam_add_tags(stream.data_ptr(), tx_tag=True, targets=names, offsets=offs, nbytes=len(hdr) + rec.size,
                                want_output=want_output)

    run(False)  # warm-up
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t1 = time.perf_counter()
    r = run(False)
    ctx.synchronize()
    wall = time.perf_counter() - t1
    prof = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith("k_bt")}
    ctx.profile(False)
    kernels_ms = sum(prof.values())
    in_bytes = n * R
    # every record gains RX:Z:<10> (14 bytes), CR:Z:<16> (20) and, when mapped, tx:Z:<15> (19)
    want_out = in_bytes + 34 * n + 19 * int((tid >= 0).sum())
    out = {"what": "bam_add_tags alignment loop (fqg_bam_add_tags), names as fastq_pre_barcodes writes them, --tx",
           "alignments": n, "record_bytes": R, "out_bytes": r["out_bytes"], "out_bytes_expected": want_out,
           "size_as_expected": r["out_bytes"] == want_out and r["n_tagged"] == n and r["code"] == 0,
           "wall_ms_one_call": wall * 1e3, "kernels_ms": kernels_ms, "kernels_ms_breakdown": prof,
           "Malignments_per_s_kernels_only": n / (kernels_ms * 1e-3) / 1e6 if kernels_ms else None,
           "algorithmic_GB": (in_bytes + r["out_bytes"]) / 1e9,
           "achieved_GBps_kernels": (in_bytes + r["out_bytes"]) / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None}
    # the first 2 000 alignments against the oracle
    from oracle import bam_tags_oracle as bto
    m = min(n, 2000)
    small = hdr + rec[:m].tobytes()
    got = ctx.bam_add_tags(small, tx_tag=True, targets=names)
    want, _ = bto.add_tags_stream(small, tx_tag=True)
    out["first_2000_alignments_identical_to_oracle"] = got["code"] == 0 and got["records"] == want[len(hdr):]
    # the reference program on a bounded sample (its BGZF inflate and deflate, single-threaded, are part of it)
    ref = os.path.join(REPO, "oracle", "_ref", "bam_add_tags")
    if os.path.exists(ref):
        ms = min(n, 500_000)
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "in.bam"), "wb") as f:
                f.write(bamgen.bgzf(hdr + rec[:ms].tobytes(), level=1))
            t2 = time.perf_counter()
            p = subprocess.run(["bam_add_tags", "--inbam", "in.bam", "--outbam", "out.bam", "--tx"], executable=ref, cwd=tmp,
                               capture_output=True)
            secs = time.perf_counter() - t2
            # the drop-in program on the same file: what it writes must inflate to what the reference writes
            mine = os.path.join(REPO, "bin", "bam_add_tags")
            if os.path.exists(mine) and p.returncode == 0:
                import gzip
                import hashlib
                t3 = time.perf_counter()
                q = subprocess.run(["bam_add_tags", "--inbam", "in.bam", "--outbam", "mine.bam", "--tx"], executable=mine, cwd=tmp,
                                   capture_output=True)
                secs_mine = time.perf_counter() - t3
                same = False
                if q.returncode == 0:
                    h = [hashlib.sha256(gzip.decompress(open(os.path.join(tmp, f), "rb").read())).hexdigest()
                         for f in ("out.bam", "mine.bam")]
                    same = h[0] == h[1] and q.stderr == p.stderr
                out["program_vs_reference_program"] = {
                    "alignments": ms, "reference_s": secs, "bin_bam_add_tags_s": secs_mine,
                    "includes": "process start, HIP initialisation, BGZF inflate and deflate (all cores here, one thread there)",
                    "inflated_output_and_stderr_identical": same}
        out["cpu_baseline"] = {"value": ms / secs / 1e6, "unit": "Malignments/s", "cores": 1, "kind": "reference",
                               "sample": f"first {ms} alignments as a BGZF file; reference bam_add_tags --tx (single-threaded, BGZF "
                                         "inflate and deflate included)", "seconds": secs, "ok": p.returncode == 0}
    return out


def umi_extra(ctx, torch, dev, n_triples):
    """bam_umi_count's alignment loop (fqg_umi_count) on BASELINE.json configs[3]: CR-sorted synthetic
    alignments, 10 k cells x 20 k genes, `n_triples` distinct (cell, gene, UMI) + 30 % duplicate reads,
    inflated records resident in HBM.  Checked at FULL size against the reference program itself
    (oracle/_ref/bam_umi_count on the same alignments as a BGZF file): the matrix lines fqg_umi_count returns
    and all six files the drop-in program writes must be identical to the reference's - including the
    (cell, gene) sets in which the reference's RL_Tree loses or invents members (fqg_rl_sim.h)."""
    import numpy as np

    from tests import bamgen  # generator only (no oracle code)

    rng = np.random.default_rng(4242)
    t0 = time.perf_counter()
    rec, cell, gene, umi = bamgen.config4(rng, n_cells=10000, n_genes=20000, n_triples=n_triples)
    hdr = bamgen.header()
    n = rec.shape[0]
    gen_s = time.perf_counter() - t0
    stream = torch.empty(len(hdr) + rec.size + 64, dtype=torch.uint8, device=dev)
    stream[: len(hdr)] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(dev)
    stream[len(hdr): len(hdr) + rec.size] = torch.from_numpy(rec.reshape(-1)).to(dev)
    torch.cuda.synchronize()
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(bamgen.REC_BYTES) + np.uint64(len(hdr)))

    def run(want_entries, **kw):
        return ctx.umi_count(stream.data_ptr(), offsets=_OffsetArray(offs), nbytes=len(hdr) + rec.size,
                             want_entries=want_entries, **kw)

    run(False)  # warm-up
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t1 = time.perf_counter()
    r = run(False)
    ctx.synchronize()
    wall = time.perf_counter() - t1
    prof = {k: v[1] / max(1, v[0]) for k, v in ctx.profile_read().items() if k.startswith(("k_umi", "k_rl"))}
    ctx.profile(False)
    got = run(True)
    # the same call with the record index resident in HBM beside the records (FQG_MEM_DEVICE_INDEXED): no 52 MB upload
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    torch.cuda.synchronize()

    def run_indexed(want_entries):
        return ctx.umi_count(stream.data_ptr(), nbytes=len(hdr) + rec.size, want_entries=want_entries,
                             offsets_device=(d_offs.data_ptr(), n))

    run_indexed(False)
    ctx.synchronize()
    t1 = time.perf_counter()
    run_indexed(False)
    ctx.synchronize()
    wall_indexed = time.perf_counter() - t1
    got_indexed = run_indexed(True)
    kernels_ms = sum(prof.values())
    out = {
        "what": "bam_umi_count alignment loop + output decisions (fqg_umi_count), BASELINE.json configs[3]",
        "wall_ms_one_call_index_in_hbm": wall_indexed * 1e3,
        "index_in_hbm_identical": got_indexed["code"] == got["code"] and got_indexed.get("entries") == got.get("entries"),
        "alignments": n, "distinct_triples": int(len(np.unique((cell.astype(np.int64) * 20000 + gene) * 4 ** 10 + umi.astype(np.int64)))),
        "cells": got["n_cells"], "genes": got["n_features"],
        "matrix_lines": len(got["entries"][0]) if got["code"] == 0 else None,
        "record_bytes": bamgen.REC_BYTES, "wall_ms_one_call_incl_allocations": wall * 1e3,
        "kernels_ms": kernels_ms, "Malignments_per_s_wall": n / wall / 1e6,
        "Malignments_per_s_kernels_only": n / (kernels_ms * 1e-3) / 1e6 if kernels_ms else None,
        "input_GBps_kernels_only": n * bamgen.REC_BYTES / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None,
        "kernels_ms_breakdown": prof,
        "rl_tree_replay": {"sets_replayed": got["rl_replayed"], "alignments_decided_differently_from_a_set": got["rl_changed"],
                           "reads_of_memory_the_reference_never_wrote": got["rl_undefined"],
                           "unresolved": got["rl_unresolved"]},
        "host_generation_s": gen_s,
    }
    # the reference program on the SAME alignments (the whole of configs[3])
    ref = os.path.join(REPO, "oracle", "_ref", "bam_umi_count")
    mine = os.path.join(REPO, "bin", "bam_umi_count")
    if os.path.exists(ref) and os.path.exists(mine):
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "in.bam"), "wb") as f:
                f.write(bamgen.bgzf(hdr + rec.tobytes(), level=1))
            files, secs = {}, {}
            for tag, exe in (("ref", ref), ("gpu", mine)):
                t2 = time.perf_counter()
                p = subprocess.run(["bam_umi_count", "--bam", "in.bam", "--ucounts", tag + "_u", "--rcounts", tag + "_r"],
                                   executable=exe, cwd=tmp, capture_output=True)
                secs[tag] = time.perf_counter() - t2
                files[tag] = [open(os.path.join(tmp, tag + b + e), "rb").read() if p.returncode == 0 else None
                              for b in ("_u", "_r") for e in ("", "_rows", "_cols")]

        def lines(blob):
            return [tuple(int(x) for x in ln.split()) for ln in blob.decode().splitlines()[2:]]
        abi_same = (files["ref"][0] is not None and got["code"] == 0 and got["entries"][0] == lines(files["ref"][0])
                    and got["entries"][1] == lines(files["ref"][3]))
        if files["ref"][0] is not None and got["code"] == 0:
            mine_u, ref_u = set(got["entries"][0]), set(lines(files["ref"][0]))
            out["matrix_lines_differing_from_reference"] = len(mine_u ^ ref_u)
            strict = run(True, strict_set=True)  # what a set would have given (the extra mode), for the record
            out["matrix_lines_a_set_would_get_differently"] = len(set(strict["entries"][0]) ^ ref_u)
        out["matrix_identical_to_reference_program"] = bool(abi_same)
        out["cpu_baseline"] = {
            "value": n / secs["ref"] / 1e6, "unit": "Malignments/s", "cores": 1, "kind": "reference",
            "sample": f"all {n} alignments of configs[3] as a BGZF level-1 file; reference bam_umi_count "
                      "(single-threaded), whole program incl. BGZF inflate and writing the files",
            "seconds": secs["ref"], "drop_in_program_seconds_same_input": secs["gpu"],
            "files_byte_identical_to_reference": files["ref"] == files["gpu"] and files["ref"][0] is not None,
        }
    return out


class _OffsetArray:
    """numpy uint64 offsets presented the way Context.umi_count takes precomputed offsets"""

    def __init__(self, a):
        import ctypes as C
        self.a = a
        self.n = int(a.size)
        self.c = (C.c_uint64 * max(1, self.n)).from_buffer(a)

    def __len__(self):
        return self.n


def kernel_sources_digest():
    """what a committed PMC traffic file is valid for: the text of the framing kernels it measured"""
    import hashlib
    h = hashlib.sha256()
    for name in ("fqg_kernels.hip", "fqg_stream_kernels.hip"):
        with open(os.path.join(REPO, "fastq_utils_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(kernel, n_reads, read_len, record_bytes):
    """HBM bytes per launch of `kernel` from the newest PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 runs, corrected as MI355X_MICROARCH.md prescribes, by
    tools/pmc_traffic.py).  Only valid for the exact workload AND kernel text it was measured on: a file that
    names other kernel sources is refused (traffic null, the reason in traffic_source)."""
    import glob
    paths = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_traffic_{n_reads // 1_000_000}M_{read_len}bp.json")))
    if not paths:
        return None, None
    path = paths[-1]
    with open(path) as f:
        t = json.load(f)
    if int(t["image_bytes"]) != n_reads * record_bytes:
        return None, None
    rel = os.path.relpath(path, REPO)
    if t.get("kernel_sources_digest") != kernel_sources_digest():
        return None, f"stale: {rel} was measured on other kernel sources"
    want = {"k_frame_fast": "k_frame_fast_t", "k_stream_redo": "k_frame_fast_t"}.get(kernel, kernel)
    by_name = {}
    for k, v in t["kernels"].items():
        by_name.setdefault(k.split("<")[0], []).append(v)
    if want not in by_name:
        return None, None
    total = sum(v["total"] for v in by_name[want]) / 1e9
    if want == "k_stream_pass1" and "k_stream_pass1_lines" in by_name:
        # the pass in parts: one k_stream_pass1 launch and launches_b / launches_a k_stream_pass1_lines launches per step
        # (the file holds averages per launch and launch counts): the figure is the pass-1 traffic of ONE STEP
        a, b = by_name["k_stream_pass1"][0], by_name["k_stream_pass1_lines"][0]
        if a.get("launches") and b.get("launches"):
            total = (a["total"] + b["total"] * b["launches"] / a["launches"]) / 1e9
    return total, rel


def flush_c_stdio():
    """A library's printf into a pipe sits in libc's buffer until the process leaves through exit(): flushed now, it
    cannot land behind the bench line (which must be the last line of stdout)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def under_a_profiler():
    """rocprofv3 preloads its tool library and writes its output when the process leaves through exit()"""
    if os.environ.get("FQGPU_BENCH_PLAIN_EXIT"):
        return True
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ)


def launch_ranks(n_gpus):
    """`python3 bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as CHILDREN (this process
    has not touched the GPU - nothing that initialises HIP is imported before this point - and never execs), relay
    what they print and leave with the launcher's status.  With FQGPU_BENCH_ONE_DEVICE=1 all ranks share GPU 0."""
    import socket

    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait()
    except BaseException:  # interrupted: take the whole group of ranks down with us
        import signal
        try:
            os.killpg(child.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        raise
    sys.exit(rc)


def main():
    a = parse()
    if a.gz_helper:
        gz_compress_file(a.gz_helper[0], a.gz_helper[1], int(a.gz_helper[2]))
        return
    if a.bgzf_helper:
        bgzf_compress_file(*a.bgzf_helper)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import fastq_utils_amd as fq  # preloads torch's HIP runtime before libfqgpu.so
    import torch
    import torch.distributed as dist

    # (test hook: FQGPU_BENCH_ONE_DEVICE=1 runs every rank on GPU 0 over gloo, to exercise the multi-rank control
    # flow on a one-GPU box; RCCL itself refuses two ranks on one device)
    one_device = os.environ.get("FQGPU_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = fq.Context(local_rank)
    R = fq.abi.synth_record_bytes(a.read_len)
    n = a.reads
    image = torch.empty(n * R, dtype=torch.uint8, device=dev)
    ctx.synth_fastq(image.data_ptr(), n, a.read_len, first_index=rank * n, seed=12345)
    ctx.synchronize()
    head = bytes(image[: 4 * R].cpu().numpy())
    st = fq.abi.probe_first_record(head, True)  # fastq_info -r sets is_pe (src/fastq_info.c:158)
    acc = ctx.accumulator()
    flags = fq.abi.VALIDATE_FORCE_EXACT if a.force_exact else 0
    if a.two_pass:
        flags |= fq.abi.VALIDATE_TWO_PASS

    def step():
        return ctx.validate(image.data_ptr(), acc, st, final=True, flags=flags, nbytes=n * R)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        res = step()
    acc.reset()
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)

    stats = acc.read()
    problem = None
    if not (res["code"] == 0 and res["n_records"] == n):
        problem = f"rank {rank}: validate returned {res}"
    elif not (stats["num_rds"] == n * a.steps and stats["min_rl"] == a.read_len + 1 == stats["max_rl"]):
        problem = f"rank {rank}: statistics {stats}"
    if world > 1:
        # a wrong result on ANY rank fails the whole job, rank 0 included (its exit code is the one that is looked at)
        bad = torch.tensor([0 if problem is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) and problem is None:
            problem = f"rank {rank}: another rank reported a wrong result"
    if problem is not None:
        print(problem, file=sys.stderr, flush=True)
        os._exit(5)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        blobs = [None] * world
        dist.all_gather_object(blobs, acc.export())
        if rank == 0:
            for b in blobs[1:]:
                acc.merge(b)
            assert acc.read()["num_rds"] == n * a.steps * world

    out = None
    if rank == 0:
        kernels = {k: v for k, v in prof.items() if k.startswith("k_") and v[0] > 0}
        dom = max(kernels, key=lambda k: kernels[k][1])
        launches, total_ms = kernels[dom]
        algo_bytes = n * (R + ALGO_BYTES_PER_READ_EXTRA)  # per step: one batch
        dom_label = dom
        if dom in ("k_stream_pass1", "k_stream_pass1_lines") and "k_stream_pass1_lines" in kernels:
            # the streaming pass in parts (round 6): pass 1 of a step is one k_stream_pass1 launch + the k_stream_pass1_lines
            # launches, which carry the line workers of the part before beside their chunks.  Priced together: every
            # launch processes its share of the step's bytes, so bytes per launch / average launch duration is the
            # step's bytes over the sum of the launches' durations.
            a1, b1 = kernels.get("k_stream_pass1", (0, 0.0)), kernels["k_stream_pass1_lines"]
            launches, total_ms = a1[0] + b1[0], a1[1] + b1[1]
            dom = "k_stream_pass1"
            dom_label = (f"k_stream_pass1 in {launches // max(1, a.steps)} launches per step: 1 x k_stream_pass1 + "
                         f"{b1[0] // max(1, a.steps)} x k_stream_pass1_lines (pass 1 of a part beside the line workers of the part before)")
        avg_ms = total_ms / launches
        bytes_per_launch = algo_bytes * a.steps / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        all_ms = sum(v[1] for v in kernels.values()) / a.steps
        traffic, traffic_src = committed_traffic(dom, n, a.read_len, R)
        out = {
            "metric": "Mreads/s validated (fastq_info, 150bp)",
            "value": n * a.steps * world / dt / 1e6,
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"HBM-resident image (host-fed rates: see host_fed): fastq_info -r (frame + validate + "
                            f"stats), {n} synthetic {a.read_len}bp SE reads per GPU, {R} B/record (BASELINE configs[1])",
                "reads_per_gpu": n, "read_len": a.read_len, "record_bytes": R, "path": res["path"],
            },
            "roofline": {
                "bound": "hbm", "kernel": dom_label, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per step (the launches of the dominant kernel)",
                "traffic_source": traffic_src,
                # the same kernel priced on the bytes the counters saw instead of SURVEY 8d's algorithmic figure (which
                # counts a 32-byte descriptor per record that a call which only validates no longer writes)
                "frac_on_measured_bytes": (traffic / (total_ms / a.steps * 1e-3) / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_step": algo_bytes, "dominant_ms_per_step": total_ms / a.steps,
                "all_kernels_ms_per_step": all_ms,
                "pipeline_achieved": algo_bytes / (all_ms * 1e-3) / 1e9,
                "pipeline_frac": algo_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # ... and by the wall clock of the timed steps (host gaps included): what the driver's clock sees
                "step_frac_wall": algo_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                "launches": launches,
                "launches_per_step": sum(v[0] for v in kernels.values()) / a.steps,
                "kernels_ms_per_step": {k: v[1] / a.steps for k, v in kernels.items()},
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            m = min(n, a.cpu_sample_reads)
            out["cpu_baseline"] = cpu_baseline(bytes(image[: m * R].cpu().numpy()), m)
        # the measured numbers go out before any untimed extra runs (marked "final": false); the line is printed again,
        # complete, as the last line of stdout
        print(headline_line(out, final=False), flush=True)
    # The extra below is the only part of a --gpus N run with collectives in the data path (RCCL all-to-all).
    # If it should hang on some topology, the line above must still come out: a watchdog prints it and ends
    # every rank.
    import threading

    def give_up():
        if rank == 0:
            out["dedup_extra"] = {"error": "timed out (watchdog) - the headline numbers above are unaffected"}
            print(headline_line(out), flush=True)
        os._exit(3)  # the headline line is out, but the run did not complete: not a success

    watchdog = None
    if not a.no_dedup_extra and world > 1:
        watchdog = threading.Timer(float(os.environ.get("FQGPU_BENCH_EXTRA_TIMEOUT", "240")), give_up)
        watchdog.daemon = True
        watchdog.start()
    dedup = None
    if not a.no_dedup_extra and (world > 1 or a.dedup_extra):
        # untimed extra on every rank: the unique-name test of fastq_info's index mode over the names of
        # ALL ranks: 16-byte (fingerprint, global index) pairs, one all-to-all over RCCL, owner
        # sets, candidates confirmed on the name bytes.  Names are unique by construction -> no finding.
        try:
            from fastq_utils_amd import dist as fdist

            own_group = False
            if world == 1 and not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29655")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                own_group = True
            rv = ctx.validate(image.data_ptr(), None, st, final=True, flags=fq.abi.VALIDATE_NO_STATS, nbytes=n * R)
            frame = ctx.retain_frame()
            barrier()
            ctx.profile(True)
            ctx.profile_reset()
            t1 = time.perf_counter()
            hit = fdist.global_first_duplicate(ctx, [(frame, rv["n_records"])], st, rank * n, device=dev)
            barrier()
            dt_dedup = time.perf_counter() - t1
            pd = ctx.profile_read()
            # ... and the file-2 loop of a pair over all ranks (BASELINE configs[4]): a DISTINCT mate file - the "2:N:0"
            # records - of which this rank holds the shard whose mates the NEXT rank holds (every name finds its mate on
            # another rank: the exchange carries real traffic)
            other = (rank + 1) % world
            image2 = torch.empty(n * R + 64, dtype=torch.uint8, device=dev)
            ctx.synth_fastq(image2.data_ptr(), n, a.read_len, first_index=other * n, seed=12345, mate=2)
            ctx.synchronize()
            st2 = fq.abi.probe_first_record(bytes(image2[: 4 * R].cpu().numpy()), True)
            rv2 = ctx.validate(image2.data_ptr(), None, st2, final=True, flags=fq.abi.VALIDATE_NO_STATS, nbytes=n * R)
            frame2 = ctx.retain_frame()
            barrier()
            t2 = time.perf_counter()
            pairing = fdist.global_pairing(ctx, [(frame, rv["n_records"])], st, rank * n, [(frame2, rv2["n_records"])], st2,
                                           other * n, device=dev)
            barrier()
            dt_pair = time.perf_counter() - t2
            # ... and what the usual placement costs - every rank holding the SAME records of both files, file 1 known to
            # be free of repeats (fastq_info knows by then): names compared by position, no exchange (one rank only: the
            # shards above are deliberately the next rank's)
            by_position = None
            if world == 1:
                t3 = time.perf_counter()
                pp = fdist.global_pairing(ctx, [(frame, rv["n_records"])], st, 0, [(frame2, rv2["n_records"])], st2, 0,
                                          device=dev, file1_unique=True)
                ctx.synchronize()
                by_position = {"wall_ms": (time.perf_counter() - t3) * 1e3, "by_position": bool(pp.get("by_position")),
                               "ok": pp["matched"] == n and pp["first_unpaired"] is None}
            ctx.profile(False)
            frame.release()
            frame2.release()
            del image2
            tt = torch.tensor([dt_dedup, dt_pair], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_pair = float(tt[1].item())
            tt = tt[:1]
            dedup = {
                "what": "unique read names over all ranks: fingerprint export + all-to-all (RCCL) + owner sets + candidate check",
                "names_total": n * world, "finding": None if hit is None else [int(hit[0]), hit[1].decode("latin-1")],
                "wall_ms_max_over_ranks": float(tt.item()) * 1e3,
                "Mnames_per_s_whole_job": n * world / float(tt.item()) / 1e6,
                "bytes_exchanged_per_rank": n * 16,
                "kernels_ms_rank0": {k: v[1] for k, v in pd.items() if k.startswith("k_") and v[0] > 0},
                "pairing": {"what": "file-2 loop over all ranks, two distinct mate files, a rank's file-2 shard holding the mates of the "
                                    "next rank's file-1 shard (both files' names exchanged, runs classified on the owners; configs[4])",
                            "pairs_total": n * world, "config4_size": bool(a.config4),
                            "matched": pairing["matched"], "leftover": pairing["leftover"], "unpaired": pairing["unpaired"],
                            "ok": pairing["matched"] == n * world and pairing["first_unpaired"] is None,
                            "wall_ms_max_over_ranks": dt_pair * 1e3,
                            "Mpairs_per_s_whole_job": n * world / dt_pair / 1e6,
                            "same_shards_by_position": by_position},
            }
            if own_group:
                dist.destroy_process_group()
        except Exception as e:  # the headline line must survive a failure of an extra
            dedup = {"error": repr(e)[:300]}

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        extras = {}

        def extra(name, fn):
            """run one untimed extra; whatever it does, the bench line still goes out; its block is a line of its own"""
            try:
                extras[name] = fn()
            except Exception as e:
                extras[name] = {"error": repr(e)[:300]}
            print(extra_line(name, extras[name]), flush=True)
            return extras[name]

        if dedup is not None:
            out["dedup_extra"] = dedup
            print(extra_line("dedup_extra", dedup), flush=True)
        if world == 1 and not a.no_index_extra:
            # untimed extra: fastq_info's default mode = the same pass + the unique read-name index, and the file-2 loop
            # of a pair (src/fastq_info.c:333-356: the same names once more as the second file - every record finds its
            # name, confirms it on the bytes and takes the entry).  Twice: with the header lines captured by the
            # streaming pass (FQG_VALIDATE_NAMES, what the programs do) and through the line index (frames that were
            # not streamed take that path).
            def default_mode(names):
                extra_flags = fq.abi.VALIDATE_NAMES if names else 0

                def index_pass(lookups):
                    acc2 = ctx.accumulator()
                    ctx.profile(True)
                    ctx.profile_reset()
                    t1 = time.perf_counter()
                    # (one file: nobody will look a name up - the streaming pass hashes the headers itself, 16-byte digests)
                    nf = (fq.abi.VALIDATE_NAME_DIGESTS if names and not lookups else extra_flags)
                    r2 = ctx.validate(image.data_ptr(), acc2, st, final=True, flags=fq.abi.VALIDATE_COUNT_TWICE | nf, nbytes=n * R)
                    idx = ctx.name_index(n)
                    if not lookups:
                        idx.expect_lookups(False)
                    ir = idx.insert_unique(st)
                    ctx.synchronize()
                    t2 = time.perf_counter()
                    p2 = ctx.profile_read()
                    ctx.profile(False)
                    acc2.close()
                    assert r2["code"] == 0 and ir["code"] == 0 and ir["n_entries"] == n, (r2, ir)
                    # (totals of this one pass: a scope with several launches - the streaming pass in parts - counts whole)
                    kern = {k: v[1] for k, v in p2.items() if k.startswith("k_") and v[0] > 0}
                    ki = kern.get("k_names_insert", kern.get("k_index_insert", 0.0)) + sum(v for k, v in kern.items() if k.startswith("k_names_build"))
                    total = sum(kern.values())
                    return idx, {
                        "wall_ms_one_pass_incl_allocations": (t2 - t1) * 1e3,
                        "kernels_ms": kern, "all_kernels_ms": total, "insert_ms": ki,
                        "ms_over_validate_only": total - all_ms,
                        "index_entries": ir["n_entries"], "names_from_capture_records": idx.names_captured(),
                        "Mreads_per_s_kernels_only": n / (total * 1e-3) / 1e6,
                        "insert_GBps_at_56B_per_name": 56.0 * n / (ki * 1e-3) / 1e9 if ki else None,
                    }

                # one file: the index is only the uniqueness test (no name records kept).  Once untimed first: a process's
                # first pass pays for the first touch of every buffer it allocates (1 - 2 ms of the kernels' time), the
                # programs' later pieces and a library user's later calls do not
                idx, _ = index_pass(False)
                idx.close()
                idx, d = index_pass(False)
                idx.close()
                # a pair: file 1 into an index that will be asked, then the same names as file 2
                try:
                    idx, d1 = index_pass(True)
                    d["pair_first_file"] = {k: d1[k] for k in ("insert_ms", "all_kernels_ms", "ms_over_validate_only")}
                    ctx.profile(True)
                    ctx.profile_reset()
                    # (as the program's file-2 loop calls it, host/fastq_info.cpp run_pair_second_file: file 1's accumulator
                    # goes on counting - with statistics wanted the image takes the pass in parts)
                    acc2 = ctx.accumulator()
                    ctx.validate(image.data_ptr(), acc2, st, final=True, flags=extra_flags, nbytes=n * R)
                    acc2.close()
                    mr = idx.match_delete(st)
                    ctx.synchronize()
                    p3 = ctx.profile_read()
                    ctx.profile(False)
                    k3 = {k: v[1] for k, v in p3.items() if k.startswith("k_") and v[0] > 0}
                    km = k3.get("k_names_match", k3.get("k_index_match_delete", 0.0))
                    d["file2_loop"] = {
                        "match_ms": km, "kernels_ms": k3, "all_kernels_ms": sum(k3.values()), "code": mr["code"],
                        "entries_left": mr["n_entries"], "ok": mr["code"] == 0 and mr["n_entries"] == 0,
                        "names_from_capture_records": idx.names_captured(),
                        "match_GBps_at_56B_per_name": 56.0 * n / (km * 1e-3) / 1e9 if km else None}
                    idx.close()
                except Exception as e:
                    d["file2_loop"] = {"error": repr(e)[:200]}
                return d

            def default_mode_both():
                dm = default_mode(True)
                dm["what"] = ("fastq_info default mode: validate + insert every read name into the GPU index (all unique), then "
                              "the same names as a second file; header lines captured by the streaming pass")
                dm["k_index_insert_ms"] = dm["insert_ms"]
                try:
                    dm["through_the_line_index"] = default_mode(False)
                except Exception as e:
                    dm["through_the_line_index"] = {"error": repr(e)[:200]}
                return dm

            extra("default_mode_extra", default_mode_both)
        if world == 1 and not a.no_e2e:
            e2e = extra("e2e", lambda: e2e_block(ctx, fq, torch, dev, image, n, R, st))
            # SURVEY 8(d) "report two rates": `value` above is the kernel-side (HBM-resident) rate; this is the
            # configs[1] placement - the same reads "uncompressed in host RAM" - through the C-ABI and through the program
            abi_leg = e2e.get("abi_pinned_host_image") or {}
            cli_leg = e2e.get("cli_fastq_info_r_tmpfs_file") or {}
            out["host_fed"] = {
                "what": "configs[1] placement: the same reads uncompressed in host RAM, fed over PCIe",
                "abi_Mreads_per_s": abi_leg.get("Mreads_per_s"), "abi_GBps": abi_leg.get("PCIe_GBps"),
                "cli_Mreads_per_s": cli_leg.get("Mreads_per_s"), "cli_GBps": cli_leg.get("GBps"),
                "ceiling": "PCIe 5 x16, 63 GB/s spec = 180 Mreads/s at this record size",
                "target_Mreads_per_s": 50.0,
            }
        if world == 1 and not a.no_filters_extra:
            extra("filters_extra", lambda: filters_extra(ctx, fq, torch, dev, image, n, R, st, a.read_len))
        if world == 1 and not (a.no_umi_extra and a.no_barcodes_extra and a.no_shapes_extra):
            del image
            torch.cuda.empty_cache()
        if world == 1 and not a.no_shapes_extra:
            extra("read_shapes_extra", lambda: shapes_extra(ctx, fq, torch, dev))
            torch.cuda.empty_cache()
        if world == 1 and not a.no_barcodes_extra:
            ctx.release_scratch()  # (the 200 M-pair leg needs nearly all of the 288 GB: what earlier legs left in the context goes)
            extra("pre_barcodes_extra", lambda: barcodes_extra(ctx, fq, torch, dev, a.barcode_pairs, programs=not a.no_e2e))
            torch.cuda.empty_cache()
        if world == 1 and not a.no_umi_extra:
            extra("umi_count_extra", lambda: umi_extra(ctx, torch, dev, a.umi_triples))
        if world == 1 and not a.no_filterpair_extra:
            extra("filterpair_extra", lambda: filterpair_extra(ctx, fq, torch, dev, a.filterpair_records, a.read_len))
            torch.cuda.empty_cache()
        if world == 1 and not a.no_tags_extra:
            extra("bam_add_tags_extra", lambda: bam_tags_extra(ctx, torch, dev, a.tags_alignments))
        extras_path = a.extras_out
        if extras_path is None and os.path.isdir(os.path.join(REPO, "gpurun_out")):
            extras_path = os.path.join(REPO, "gpurun_out", "bench_extras.json")
        if extras_path:
            try:
                with open(extras_path, "w") as f:
                    json.dump(_plain({"headline": json.loads(headline_line(out, extras)), "roofline_full": out["roofline"],
                                      "cpu_baseline_full": out.get("cpu_baseline"), "dedup_extra": dedup, **extras}, sig=9), f)
                    f.write("\n")
            except OSError as e:
                print(f"bench.py: could not write {extras_path}: {e}", file=sys.stderr)
        flush_c_stdio()  # (what a library printed through C stdio - RCCL's start-up banner - goes out BEFORE the line)
        print(headline_line(out, extras), flush=True)  # the bench line: LAST on stdout
    if world > 1:
        # every rank has what it needs; a rank that failed in the extra must not keep the others waiting in a
        # collective tear-down
        sys.stdout.flush()
        os._exit(4 if (dedup is not None and "error" in dedup) else 0)
    acc.close()
    ctx.close()
    # (the line is out and everything is closed: leave without the interpreter's and the HIP runtime's tear-down - a
    # drop-in program that left through exit() ended with a segmentation fault once in a few hundred runs)
    sys.stdout.flush()
    sys.stderr.flush()
    if under_a_profiler():
        return  # (rocprofv3 writes its files from an exit hook: leave the ordinary way)
    os._exit(0)


if __name__ == "__main__":
    main()
