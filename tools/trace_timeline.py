"""Timeline of one SPD inverse from a rocprofv3 --kernel-trace CSV: per pivot group, when each chain kernel started and
ended relative to the big update launches (stream / queue per kernel).

    python tools/trace_timeline.py <kernel_trace.csv> [--family K] [--groups a:b]
"""
import argparse
import csv
import re


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n


ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--family", type=int, default=-1, help="which inverse of the trace (default: last)")
ap.add_argument("--groups", default="10:13")
args = ap.parse_args()
rows = []
with open(args.csv, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], r["Stream_Id"],
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]))))
rows.sort()
# split into families at k_hamming
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_hamming")]
fam = starts[args.family]
end = starts[args.family + 1] if args.family + 1 < len(starts) and args.family != -1 else len(rows)
fr = rows[fam:end]
inv = [r for r in fr if r[2].startswith(("k_pivot", "k_group", "k_tile_jobs", "k_gather", "k_scatter"))]
t0 = inv[0][0]
big = [r for r in inv if r[2].startswith("k_group_update<")]
print("inverse: first kernel -> last kernel end: %.3f ms; big updates: %d, sum %.3f ms, span %.3f ms" %
      ((max(r[1] for r in inv) - t0) / 1e6, len(big), sum(r[1] - r[0] for r in big) / 1e6, (big[-1][1] - big[0][0]) / 1e6))
gaps = [(big[i + 1][0] - big[i][1]) / 1e3 for i in range(len(big) - 1)]
print("gaps between consecutive big updates (us): " + " ".join("%.0f" % g for g in gaps))
a, b = (int(x) for x in args.groups.split(":"))
lo, hi = big[a][0], big[b][1]
print("kernels overlapping big updates %d..%d (time in us from the start of update %d):" % (a, b, a))
for r in inv:
    if r[1] >= lo and r[0] <= hi:
        print("  %9.1f -> %9.1f  (%7.1f)  q%s s%s  wgs %5d  %s" % ((r[0] - lo) / 1e3, (r[1] - lo) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[2]))
