"""A/B on one box: the streaming pass with name records / with name digests, and the insert behind each (CAS pass / table
built in LDS).  python tools/names_quick.py [reads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastq_utils_amd as fq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with fq.Context(0) as ctx:
    R = fq.abi.synth_record_bytes(150)
    image = torch.empty(n * R + 64, dtype=torch.uint8, device="cuda:0")
    ctx.synth_fastq(image.data_ptr(), n, 150)
    ctx.synchronize()
    head = bytes(image[:4096].cpu().numpy())
    st = fq.abi.probe_first_record(head, False)
    for rep in range(3):
        for label, flags, lookups, build in (("plain", 0, None, None), ("records", fq.abi.VALIDATE_NAMES, False, None),
                                             ("digests+cas", fq.abi.VALIDATE_NAME_DIGESTS, False, "0"),
                                             ("digests+lds", fq.abi.VALIDATE_NAME_DIGESTS, False, "1"),
                                             ("records+names", fq.abi.VALIDATE_NAMES, True, None)):
            if build is not None:
                os.environ["FQGPU_NAMES_BUILD"] = build
            else:
                os.environ.pop("FQGPU_NAMES_BUILD", None)
            acc = ctx.accumulator()
            ctx.profile(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            r = ctx.validate(image.data_ptr(), acc, st, final=True, flags=fq.abi.VALIDATE_COUNT_TWICE | flags, nbytes=n * R)
            ir = None
            if lookups is not None:
                idx = ctx.name_index(n)
                if not lookups:
                    idx.expect_lookups(False)
                ir = idx.insert_unique(st)
            ctx.synchronize()
            t1 = time.perf_counter()
            p = ctx.profile_read()
            ctx.profile(False)
            kern = {k: round(v[1] / max(1, v[0]), 3) for k, v in p.items() if k.startswith("k_") and v[0] > 0}
            keep = {k: v for k, v in kern.items() if "pass1" in k or "names" in k or "lines" in k or "index" in k}
            print(rep, label, "total_kernels_ms=%.3f" % sum(kern.values()), keep,
                  None if ir is None else (ir["code"], ir["n_entries"]), "wall_ms=%.1f" % ((t1 - t0) * 1e3), flush=True)
            if lookups is not None:
                idx.close()
            acc.close()
