#!/usr/bin/env python3
"""How the batcher's copy kernels share the link, from a rocprofv3 --kernel-trace database of a trait-level run
(tools/decoders_bench): per direction (batch_copy_kernel<*, 0> = gather, <*, 1> = scatter) the launches, their durations and the
gaps between consecutive launches; the fraction of the time with (gathers, scatters) = (g, s) kernels running at once; and the
synthesis kernels beside them.  The first `skip` of the run (warm-up) is left out.

    python tools/copy_timeline.py <results.db> [skip_fraction]"""
import sqlite3
import sys


def main(path, skip=0.4):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    if not rows:
        print("no kernels")
        return
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= lo]
    span = (t1 - lo) / 1e3
    print("window: last %.0f %% of the run, %.1f ms, %d kernel launches" % (100 * (1 - skip), span / 1e3, len(rows)))

    def kind(name):
        if "batch_copy_kernel" in name:
            return "scatter" if ", 1>" in name else "gather"
        if "batch_flag_kernel" in name:
            return "flag"
        return "synth"
    by = {}
    for name, s, e, grid in rows:
        by.setdefault(kind(name), []).append((s, e, grid))
    for k in ("gather", "scatter", "synth", "flag"):
        v = by.get(k, [])
        if not v:
            continue
        d = sorted((e - s) / 1e3 for s, e, _ in v)
        busy = sum(d)
        print("%-8s %5d launches  duration us: median %.1f  p10 %.1f  p90 %.1f  sum %.1f ms (%.0f %% of the window if laid end to end)  median grid %d" % (
            k, len(v), d[len(d) // 2], d[len(d) // 10], d[len(d) * 9 // 10], busy / 1e3, 100 * busy / span, sorted(g for _, _, g in v)[len(v) // 2]))
    # concurrency by direction
    ev = []
    for k in ("gather", "scatter"):
        for s, e, _ in by.get(k, []):
            ev.append((s, 1, k))
            ev.append((e, -1, k))
    ev.sort()
    cur = {"gather": 0, "scatter": 0}
    last = lo
    share = {}
    for t, d, k in ev:
        share[(cur["gather"], cur["scatter"])] = share.get((cur["gather"], cur["scatter"]), 0) + (t - last)
        cur[k] += d
        last = t
    share[(0, 0)] = share.get((0, 0), 0) + (t1 - last)
    tot = sum(share.values()) or 1
    print("(gathers, scatters) running at once -> share of the window:")
    for key in sorted(share):
        if share[key] / tot >= 0.005:
            print("   %s  %.3f" % (key, share[key] / tot))
    both = sum(v for (g, s), v in share.items() if g and s) / tot
    one = sum(v for (g, s), v in share.items() if bool(g) != bool(s)) / tot
    print("both directions busy %.3f, one direction only %.3f, neither %.3f" % (both, one, 1 - both - one))


def dump(path, skip=0.6, count=80):
    """the launches one after the other: kind, queue, start and end relative to the first (us)"""
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
    st = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else "0")
    rows = c.execute("select name, start, end, grid_x, %s, %s from kernels order by start" % (q, st)).fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= lo][:count]
    print("columns of `kernels`:", ", ".join(cols))
    base = rows[0][1]
    for name, s, e, grid, qq, ss in rows:
        k = ("scatter" if ", 1>" in name else "gather") if "batch_copy_kernel" in name else ("flag" if "batch_flag" in name else name.split("(")[0].split("::")[-1][:28])
        print("%-28s queue %-4s stream %-4s start %9.1f  end %9.1f  (%.1f us)  grid %d" % (k, qq, ss, (s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, grid))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "dump":
        dump(sys.argv[1])
    else:
        main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
