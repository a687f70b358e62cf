"""Summaries of rocprofv3 CSV output for profiles/ (the numbers bench.py's roofline object and DESIGN.md quote).

    python tools/prof_summary.py pmc <dir with *_counter_collection.csv> ... --out profiles/rNN_pmc.json
    python tools/prof_summary.py traffic <FETCH dir> <WRITE dir> --out profiles/rNN_pmc_update_traffic.json

`pmc`: per kernel and counter, launches / mean / sum (counters as reported; FETCH_SIZE and WRITE_SIZE are in KB).
    python tools/prof_summary.py calib <FETCH dir> <WRITE dir> --out profiles/rNN_fetch_calibration.json

`traffic`: HBM bytes per launch of the dominant kernel (k_sweep: the whole SPD inverse as one persistent launch) from two
separate passes, one counter each, as MI355X_MICROARCH.md prescribes, corrected with the factors of `calib`.
`calib`: counter / true-bytes factors from tools/ubench_fetch_calib (copy kernels with the sweep kernel's access mix).
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
    launches_of = {}
    for d, cname in ((args.fetch, "FETCH_SIZE"), (args.write, "WRITE_SIZE")):
        vals, allk = [], defaultdict(float)
        nl = defaultdict(int)
        launches_of[cname] = nl
        for r in read_counters(d):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            allk[k] += float(r["Counter_Value"])
            nl[k] += 1
            if k.startswith("k_sweep") and "prep" not in k:
                vals.append((float(r["Counter_Value"]), k))
        res[cname] = vals
        per_kernel[cname + "_all_kernels_sum_kb"] = dict(allk)
    out = {"kernel": "k_sweep: the SPD inverse as one persistent launch",
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                      "--steps 1 --warmup 1 --no-cpu-baseline (two separate passes)",
           "workload": {"N": args.N, "M": args.M, "score": args.score}}
    n = min(len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]))
    f = sum(v for v, _ in res["FETCH_SIZE"]) / max(1, len(res["FETCH_SIZE"])) * 1024.0
    w = sum(v for v, _ in res["WRITE_SIZE"]) / max(1, len(res["WRITE_SIZE"])) * 1024.0
    cf, cw = 1.0, 1.0
    if args.calib:
        cal = json.load(open(args.calib))
        cf, cw = cal["k_mix"]["fetch_counter_over_bytes"], cal["k_mix"]["write_counter_over_bytes"]
        out["calibration"] = {"file": os.path.basename(args.calib), "fetch_counter_over_bytes": cf, "write_counter_over_bytes": cw}
    # every kernel of the run, per launch, same corrections (bench.py fills the `traffic` of its roofline_hbm records from this)
    pk = {}
    for k in per_kernel["FETCH_SIZE_all_kernels_sum_kb"]:
        nf, nw = launches_of["FETCH_SIZE"].get(k, 0), launches_of["WRITE_SIZE"].get(k, 0)
        if nf and nw:
            fb = per_kernel["FETCH_SIZE_all_kernels_sum_kb"][k] * 1024.0 / nf / cf
            wb = per_kernel["WRITE_SIZE_all_kernels_sum_kb"].get(k, 0.0) * 1024.0 / nw / cw
            pk[k] = {"launches": min(nf, nw), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    out["per_kernel"] = pk
    out.update(launches=n, fetch_counter_bytes_per_launch=f, write_counter_bytes_per_launch=w,
               fetch_bytes_per_launch=f / cf, write_bytes_per_launch=w / cw, hbm_bytes_per_launch=f / cf + w / cw,
               kernels=sorted({k for _, k in res["FETCH_SIZE"]}), raw=per_kernel)
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print("wrote", args.out, "launches", n, "fetch MB %.1f write MB %.1f" % (f / 1e6, w / 1e6))


def cmd_calib(args):
    """counter / bytes for the three kernels of tools/ubench_fetch_calib (last launch of each)"""
    true = {"k_copy8": (2 ** 28 * 8, 2 ** 28 * 8), "k_copy16": (2 ** 28 * 8, 2 ** 28 * 4),
            "k_mix": (4096 * 16384 * 8 * 7, 4096 * 16384 * 8)}
    out = {"tool": "tools/ubench_fetch_calib.hip", "unit": "FETCH_SIZE / WRITE_SIZE are reported in KB (x 1024 here)"}
    got = {}
    for d, cname in ((args.fetch, "FETCH_SIZE"), (args.write, "WRITE_SIZE")):
        for r in read_counters(d):
            if r["Counter_Name"] == cname:
                got.setdefault(short(r["Kernel_Name"]), {})[cname] = float(r["Counter_Value"]) * 1024.0  # last launch wins
    for k, (rd, wr) in true.items():
        g = got.get(k, {})
        out[k] = {"true_read_bytes": rd, "true_write_bytes": wr, "FETCH_SIZE_bytes": g.get("FETCH_SIZE"),
                  "WRITE_SIZE_bytes": g.get("WRITE_SIZE"),
                  "fetch_counter_over_bytes": g.get("FETCH_SIZE", 0.0) / rd, "write_counter_over_bytes": g.get("WRITE_SIZE", 0.0) / wr}
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (v["fetch_counter_over_bytes"], v["write_counter_over_bytes"]) for k, v in out.items() if k.startswith("k_")}))


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
    b.add_argument("--calib", default=None)
    b.add_argument("--N", type=int, default=500)
    b.add_argument("--M", type=int, default=50000)
    b.add_argument("--score", default="frob")
    b.set_defaults(fn=cmd_traffic)
    c = sub.add_parser("calib")
    c.add_argument("fetch")
    c.add_argument("write")
    c.add_argument("--out", required=True)
    c.set_defaults(fn=cmd_calib)
    args = ap.parse_args()
    args.fn(args)
