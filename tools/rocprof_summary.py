#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (rocpd sqlite) as plain text for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x.kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s} "
          f"{'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>10s} {'wg':>4s}")
    for name, n, tot, avg, mn, mx, vg, sg, lds, gx, wx in rows:
        short = name.split("(")[0][-60:]
        print(f"{short:60s} {n:6d} {tot/1e6:10.3f} {avg/1e3:11.2f} {mn/1e3:11.2f} {mx/1e3:11.2f} "
              f"{100*tot/total:6.2f} {vg or 0:5d} {sg or 0:5d} {lds or 0:6d} {gx or 0:10d} {wx or 0:4d}")
    if "--pmc" in sys.argv:
        try:
            q = ("select k.name, p.name, count(*), sum(e.value), avg(e.value) from pmc_events e ... ")
        except Exception:
            pass


if __name__ == "__main__":
    main(sys.argv[1])
