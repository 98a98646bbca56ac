#!/usr/bin/env python3
"""bench.py - Mreads/s validated by the fastq_info hot path on MI355X.

A "step" is one pass of the hot path (framing + validation + statistics, i.e. what
`fastq_info -r` does per file) over one HBM-resident batch of synthetic 150 bp reads
(BASELINE.json configs[1]: 100 M reads, 349 B/record = 34.9 GB per GPU).  Input bytes are
already in HBM when the timed region starts.  With --gpus N every rank validates its own shard
of the same size (weak scaling; no data-path collective - records are independent), and the
per-rank statistics are merged once at the end.

Prints ONE JSON line on rank 0 (see the driver contract in the task description), extended with
  roofline      HBM roofline of the dominant kernel, measured with hipEvents on the launch stream
  cpu_baseline  the reference's own fastq_info -r (oracle/_ref) timed on this box's host cores
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_READ_EXTRA = 32  # descriptor bytes per record


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample-reads", type=int, default=8_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exact", action="store_true", help="use only the wave-per-record validator")
    ap.add_argument("--two-pass", action="store_true", help="census + framing passes instead of the single-pass path")
    ap.add_argument("--no-index-extra", action="store_true",
                    help="skip the untimed extra: fastq_info default mode (validate + unique read-name index)")
    return ap.parse_args()


def cpu_baseline(image_prefix_bytes, n_reads):
    """Time the CPU side on a bounded sample of the same workload (rank 0, N=1 only).
    Prefers the reference program itself (oracle/_ref/fastq_info -r); falls back to the C
    restatement (oracle/liboracle_fq.so)."""
    from oracle import loader as orc  # the only place bench.py touches oracle/: as the baseline

    ref = os.path.join(orc.REF_DIR, "fastq_info")
    sample = f"first {n_reads} reads of the same synthetic batch ({len(image_prefix_bytes)/1e9:.2f} GB, uncompressed)"
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "sample.fastq")
            with open(path, "wb") as f:
                f.write(image_prefix_bytes)
            with open(path, "rb") as f:  # page cache warm
                while f.read(1 << 26):
                    pass
            t0 = time.perf_counter()
            p = subprocess.run([ref, "-r", path], capture_output=True)
            dt = time.perf_counter() - t0
            ok = p.returncode == 0 and (f"Number of reads: {n_reads}".encode() in p.stderr)
        return {"value": n_reads / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "reference",
                "sample": sample + "; reference fastq_info -r, 1 process (the reference is single-threaded)",
                "seconds": dt, "ok": bool(ok)}
    t0 = time.perf_counter()
    r = orc.fastq_info(image_prefix_bytes, "sample.fastq", flags=orc.FLAG_R)
    dt = time.perf_counter() - t0
    return {"value": n_reads / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": sample + "; oracle/fq_oracle.c restatement", "seconds": dt, "ok": r["exit"] == 0}


def committed_traffic(kernel, n_reads, read_len, record_bytes):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 runs, corrected as MI355X_MICROARCH.md prescribes, by
    tools/pmc_traffic.py).  Only valid for the exact workload it was measured on; otherwise None."""
    path = os.path.join(REPO, "profiles", f"r01c_traffic_{n_reads // 1_000_000}M_{read_len}bp.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    if int(t["image_bytes"]) != n_reads * record_bytes:
        return None, None
    for k, v in t["kernels"].items():
        if k.split("<")[0] == {"k_frame_fast": "k_frame_fast_t", "k_stream_redo": "k_frame_fast_t"}.get(kernel, kernel):
            return v["total"] / 1e9, os.path.relpath(path, REPO)
    return None, None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import fastq_utils_amd as fq  # preloads torch's HIP runtime before libfqgpu.so
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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

    assert res["code"] == 0 and res["n_records"] == n, res
    stats = acc.read()
    assert stats["num_rds"] == n * a.steps and stats["min_rl"] == a.read_len + 1 == stats["max_rl"], stats

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

    if rank == 0:
        kernels = {k: v for k, v in prof.items() if k.startswith("k_") and v[0] > 0}
        dom = max(kernels, key=lambda k: kernels[k][1])
        launches, total_ms = kernels[dom]
        avg_ms = total_ms / launches
        algo_bytes = n * (R + ALGO_BYTES_PER_READ_EXTRA)  # per launch: one batch
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
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
                "workload": f"fastq_info -r (frame + validate + stats) on {n} synthetic {a.read_len}bp single-end "
                            f"reads per GPU, {R} B/record, uncompressed, HBM-resident (BASELINE.json configs[1])",
                "reads_per_gpu": n, "read_len": a.read_len, "record_bytes": R, "path": res["path"],
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": avg_ms,
                "all_kernels_ms_per_step": all_ms,
                "pipeline_achieved": algo_bytes / (all_ms * 1e-3) / 1e9,
                "kernels_ms_per_step": {k: v[1] / a.steps for k, v in kernels.items()},
            },
        }
        if world == 1 and not a.no_index_extra:
            # untimed extra: fastq_info's default mode = the same pass + the unique read-name index
            acc2 = ctx.accumulator()
            ctx.profile(True)
            ctx.profile_reset()
            t1 = time.perf_counter()
            r2 = ctx.validate(image.data_ptr(), acc2, st, final=True, flags=fq.abi.VALIDATE_COUNT_TWICE, nbytes=n * R)
            idx = ctx.name_index(n)
            ir = idx.insert_unique(st)
            ctx.synchronize()
            t2 = time.perf_counter()
            p2 = ctx.profile_read()
            ctx.profile(False)
            assert r2["code"] == 0 and ir["code"] == 0 and ir["n_entries"] == n, (r2, ir)
            ki = p2.get("k_index_insert", (1, 0.0))
            out["default_mode_extra"] = {
                "what": "fastq_info default mode: validate + insert every read name into the GPU index (all unique)",
                "wall_ms_one_pass_incl_allocations": (t2 - t1) * 1e3,
                "k_index_insert_ms": ki[1] / max(1, ki[0]),
                "index_entries": ir["n_entries"],
                "Mreads_per_s_kernels_only": n / ((all_ms + ki[1] / max(1, ki[0])) * 1e-3) / 1e6,
            }
            idx.close()
            acc2.close()
        if world == 1 and not a.no_cpu_baseline:
            m = min(n, a.cpu_sample_reads)
            out["cpu_baseline"] = cpu_baseline(bytes(image[: m * R].cpu().numpy()), m)
        print(json.dumps(out), flush=True)
    acc.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
