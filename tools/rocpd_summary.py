#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs as text: per-kernel stats and per-kernel PMC averages.  Launches of one kernel
are grouped by grid size as well, so that the full-batch launches of a timed region are not averaged with the small
chunked launches of, e.g., the bench's host-to-host leg.

    python tools/rocpd_summary.py gpurun_out/prof_r01_aac/aac_results.db [...more .db] > profiles/r01_aac.txt
"""
import sqlite3
import sys


def short(name):
    name = name.replace("symaccel::(anonymous namespace)::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main(paths):
    for p in paths:
        c = sqlite3.connect(p)
        print("== %s" % p)
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                         "max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                         "max(lds_size), max(scratch_size) from kernels group by name, grid_x order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        if rows:
            print("%-90s %6s %12s %12s %12s %12s %6s %10s %5s %5s %5s %5s %7s %7s" % (
                "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid", "wg", "vgpr", "agpr", "sgpr", "lds", "scratch"))
        for r in rows:
            print("%-90s %6d %12.1f %12.2f %12.2f %12.2f %6.2f %10d %5d %5d %5d %5d %7d %7d" % (
                short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, *r[6:]))
            if r[4] > 0 and r[5] / r[4] > 3.0:
                # two populations under one name and grid (e.g. a bench's full-batch steps and the chunk launches of its
                # host-to-host leg, which pick shorter segments for the same number of wavefronts): the long one on its own
                d = [x[0] for x in c.execute("select duration from kernels where name = ? and grid_x = ?", (r[0], r[6]))]
                big = [x for x in d if x * 2 >= r[5]]
                print("%-90s %6d %12.1f %12.2f %12.2f %12.2f" % ("    of which within 2x of the longest launch", len(big), sum(big) / 1e3,
                                                               sum(big) / len(big) / 1e3, min(big) / 1e3, max(big) / 1e3))
        pm = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                       "from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
        if pm:
            print("-- PMC (per dispatch)")
            print("%-90s %-16s %6s %16s %16s %16s" % ("kernel", "counter", "n", "avg", "min", "max"))
            for r in pm:
                print("%-90s %-16s %6d %16.3f %16.3f %16.3f" % (short(r[0]), *r[1:]))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
