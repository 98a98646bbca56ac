"""Where does the time of ONE call go that is not kernel time?  Reads a rocprofv3 --kernel-trace CSV, takes the span
from the LAST launch of kernel FIRST to the next launch of kernel LAST behind it, and prints every launch in it (start
offset, duration, the idle gap in front of it) and the totals.
  python3 tools/trace_gaps.py kernel_trace.csv k_umi_parse k_umi_compact"""
import csv
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("fqg::", "").replace("void ", "")
    if "rocprim" in n:
        for key in ("onesweep", "block_sort", "merge", "histogram", "scan", "partition"):
            if key in name:
                return "rocprim:" + key
        return "rocprim"
    return n[:48]


def main():
    path, first, last = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if short(r[2]) == first]
    if not starts:
        sys.exit(f"no launch of {first}")
    a = starts[-1]
    b = next((i for i in range(len(rows) - 1, a, -1) if short(rows[i][2]) == last), len(rows) - 1)
    span = rows[a:b + 1]
    t0, busy, prev_end = span[0][0], 0, span[0][0]
    gaps = []
    for s, e, name in span:
        gap = max(0, s - prev_end)
        gaps.append(gap)
        busy += e - max(s, prev_end) if e > prev_end else 0
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f} us  {short(name)}")
        prev_end = max(prev_end, e)
    total = prev_end - t0
    print(f"launches {len(span)}, span {total / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(total - busy) / 1e3:.1f} us; "
          f"gaps over 20 us: {sum(1 for g in gaps if g > 20000)} (sum {sum(g for g in gaps if g > 20000) / 1e3:.1f} us)")


if __name__ == "__main__":
    main()
