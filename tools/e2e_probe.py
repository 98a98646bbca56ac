"""Exploration (not part of the bench contract): host-fed rates of the validate path on one GPU.
  (1) pinned host image -> fqg_validate(FQG_MEM_HOST) piece by piece (H2D + kernels, nothing to read)
  (2) bin/fastq_info -r on a tmpfs file (process start, pinned ring, parallel pread, H2D, kernels)"""
import argparse, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastq_utils_amd as fq

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100_000_000)
ap.add_argument("--piece-mb", type=int, default=1024)
ap.add_argument("--path", default="/dev/shm/fqg_e2e.fastq")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = fq.Context(0)
R = fq.abi.synth_record_bytes(150)
n = a.reads
t0 = time.perf_counter()
host = torch.empty(n * R, dtype=torch.uint8, pin_memory=True)
t_pin = time.perf_counter() - t0
step = 8_000_000
t0 = time.perf_counter()
for i in range(0, n, step):
    m = min(step, n - i)
    img = torch.empty(m * R, dtype=torch.uint8, device=dev)
    ctx.synth_fastq(img.data_ptr(), m, 150, first_index=i, seed=12345)
    ctx.synchronize()
    host[i * R:(i + m) * R].copy_(img)
torch.cuda.synchronize()
t_gen = time.perf_counter() - t0
st = fq.abi.probe_first_record(bytes(host[:4 * R].numpy()), True)
out = {"reads": n, "bytes": n * R, "pin_s": t_pin, "gen_s": t_gen}
piece = (a.piece_mb << 20) // R * R
for rep in range(2):
    acc = ctx.accumulator()
    t0 = time.perf_counter()
    done = 0
    for off in range(0, n * R, piece):
        nb = min(piece, n * R - off)
        r = ctx.validate(host.data_ptr() + off, acc, st, final=(off + nb == n * R), nbytes=nb, mem=fq.abi.MEM_HOST)
        assert r["code"] == 0, r
        done += r["n_records"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    assert done == n
    out["abi_host_fed_rep%d" % rep] = {"s": dt, "Mreads_per_s": n / dt / 1e6, "GBps": n * R / dt / 1e9}
    acc.close()
t0 = time.perf_counter()
with open(a.path, "wb") as f:
    arr = host.numpy()
    for off in range(0, n * R, 1 << 30):
        f.write(arr[off:off + (1 << 30)].data)
out["write_tmpfs_s"] = time.perf_counter() - t0
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "fastq_info")
for mb in (256, 1024):
    for thr in (8, 16, 32):
        env = dict(os.environ, FQGPU_CHUNK_MB=str(mb), FQGPU_HOST_THREADS=str(thr))
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-r", a.path], capture_output=True, env=env)
        dt = time.perf_counter() - t0
        out["cli_chunk%d_thr%d" % (mb, thr)] = {"s": dt, "Mreads_per_s": n / dt / 1e6, "rc": p.returncode,
                                               "tail": p.stdout.decode()[-200:] if p.returncode else p.stdout.decode()[-120:]}
os.unlink(a.path)
print(json.dumps(out, indent=1))
