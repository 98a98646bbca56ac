#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only)
into per-kernel HBM bytes per launch, with the corrections of MI355X_MICROARCH.md (HBM section):
both counters are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
read stream.  The factor is checked here on k_count_nl, which reads every byte of the image
exactly once with 16-byte-per-lane loads (calibration on a known byte count, as the guide asks).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
           <image_bytes> [<fetch csv of a run that contains k_count_nl>] > profiles/rNN_traffic.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fqg::", "")
            agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, write, image_bytes = sys.argv[1], sys.argv[2], float(sys.argv[3])
    (f, fn), (w, _) = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    fc = per_kernel(sys.argv[4], "FETCH_SIZE")[0] if len(sys.argv) > 4 else f
    calib = image_bytes / (fc["k_count_nl"] * 1024.0) if "k_count_nl" in fc else None  # (None: no calibration kernel in these runs)
    import hashlib
    import os
    h = hashlib.sha256()
    for name in ("fqg_kernels.hip", "fqg_stream_kernels.hip"):  # (what bench.py checks before quoting this file)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fastq_utils_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    out = {"image_bytes": image_bytes, "kernel_sources_digest": h.hexdigest()[:16],
           "fetch_correction_measured_on_k_count_nl": calib,
           "fetch_correction_applied": 2.0, "unit": "bytes per launch", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        rd = f.get(k, 0.0) * 1024.0 * 2.0
        wr = w.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"read": rd, "write": wr, "total": rd + wr, "launches": fn.get(k, 0)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
