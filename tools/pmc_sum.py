#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per kernel over every counter_collection csv below a directory.

    python tools/pmc_sum.py <dir> [<kernel name filter>]  ->  JSON {kernel: {counter: sum, "launches": n}}
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fqg::", "")
            if flt and flt not in name:
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add((path, r.get("Dispatch_Id")))
    out = {k: dict(v, launches=len(launches[k])) for k, v in agg.items()}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
