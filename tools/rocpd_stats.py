"""Kernel statistics out of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`): per kernel name calls, total,
average, min, max in microseconds -- and, with --timeline A B, every dispatch between the A-th and B-th launch of the sweep kernel.
    python tools/rocpd_stats.py results.db [--top 40] [--timeline 3 4]"""
import argparse
import re
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--timeline", type=int, nargs=2)
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, workgroup_x, stream_id from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(.*", "", n.replace("void ", ""))
    n = re.sub(r"BatchArgs<[^>]*>", "", n)
    return n[:46]


agg = {}
for n, s, e, gx, gy, wx, st in rows:
    d = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
    d[0] += 1
    d[1] += e - s
    d[2] = min(d[2], e - s)
    d[3] = max(d[3], e - s)
tot = sum(d[1] for d in agg.values())
print("%-46s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total us", "avg us", "min us", "max us", "%"))
for n, d in sorted(agg.items(), key=lambda x: -x[1][1])[: a.top]:
    print("%-46s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n, d[0], d[1] / 1e3, d[1] / d[0] / 1e3, d[2] / 1e3, d[3] / 1e3, 100.0 * d[1] / tot))
print("span of all dispatches %.3f ms, sum of kernel times %.3f ms, dispatches %d" % ((rows[-1][2] - rows[0][1]) / 1e6, tot / 1e6, len(rows)))
if a.timeline:
    sweeps = [i for i, r in enumerate(rows) if "k_sweep" in r[0] and "prep" not in r[0]]
    lo, hi = sweeps[a.timeline[0]], sweeps[a.timeline[1]]
    t0 = rows[lo][1]
    for n, s, e, gx, gy, wx, st in rows[lo : hi + 1]:
        print("%10.1f us  +%9.1f us  stream %-3s grid %8d x %-5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, gx // max(wx, 1), gy, short(n)))
