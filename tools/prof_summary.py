"""Summaries of rocprofv3 CSV output for profiles/ (the numbers bench.py's roofline object and DESIGN.md quote).

    python tools/prof_summary.py pmc <dir with *_counter_collection.csv> ... --out profiles/rNN_pmc.json
    python tools/prof_summary.py traffic <FETCH dir> <WRITE dir> --out profiles/rNN_pmc_update_traffic.json

`pmc`: per kernel and counter, launches / mean / sum (counters as reported; FETCH_SIZE and WRITE_SIZE are in KB).
`traffic`: HBM bytes per launch of the dominant kernel (the big trailing-update launches of the SPD inverse:
k_group_update / k_sweep_update with a grid of more than 1000 workgroups) from two separate passes, one counter each, as
MI355X_MICROARCH.md prescribes; no x2 correction on FETCH_SIZE (8-byte-per-lane tile loads).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def read_counters(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            rows.extend(csv.DictReader(fh))
    return rows


def cmd_pmc(args):
    acc = defaultdict(lambda: defaultdict(list))
    for d in args.dirs:
        for r in read_counters(d):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
           for k, cs in acc.items()}
    for k, cs in out.items():
        if "SQ_INSTS_MFMA" in cs and cs["SQ_INSTS_MFMA"]["sum"] > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            cs["derived"] = {"mfma_busy_cycles_per_mfma": cs["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / cs["SQ_INSTS_MFMA"]["sum"]}
            if "SQ_BUSY_CYCLES" in cs and cs["SQ_BUSY_CYCLES"]["sum"] > 0:
                cs["derived"]["mfma_busy_over_sq_busy"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / cs["SQ_BUSY_CYCLES"]["sum"]
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print("wrote", args.out, "kernels:", len(out))


def cmd_traffic(args):
    res = {}
    per_kernel = {}
    for d, cname in ((args.fetch, "FETCH_SIZE"), (args.write, "WRITE_SIZE")):
        vals, allk = [], defaultdict(float)
        for r in read_counters(d):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            allk[k] += float(r["Counter_Value"])
            if k.startswith(("k_sweep_update", "k_group_update")) and int(r["Grid_Size"]) > 1000 * int(r["Workgroup_Size"]):
                vals.append((float(r["Counter_Value"]), k))
        res[cname] = vals
        per_kernel[cname + "_all_kernels_sum_kb"] = dict(allk)
    out = {"kernel": "trailing update of the SPD inverse (k_group_update / k_sweep_update launches with more than 1000 workgroups)",
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                      "--steps 1 --warmup 1 --no-cpu-baseline (two separate passes)"}
    n = min(len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]))
    f = sum(v for v, _ in res["FETCH_SIZE"]) / max(1, len(res["FETCH_SIZE"])) * 1024.0
    w = sum(v for v, _ in res["WRITE_SIZE"]) / max(1, len(res["WRITE_SIZE"])) * 1024.0
    out.update(launches=n, fetch_bytes_per_launch=f, write_bytes_per_launch=w, hbm_bytes_per_launch=f + w,
               kernels=sorted({k for _, k in res["FETCH_SIZE"]}), raw=per_kernel)
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print("wrote", args.out, "launches", n, "fetch MB %.1f write MB %.1f" % (f / 1e6, w / 1e6))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("pmc")
    a.add_argument("dirs", nargs="+")
    a.add_argument("--out", required=True)
    a.set_defaults(fn=cmd_pmc)
    b = sub.add_parser("traffic")
    b.add_argument("fetch")
    b.add_argument("write")
    b.add_argument("--out", required=True)
    b.set_defaults(fn=cmd_traffic)
    args = ap.parse_args()
    args.fn(args)
