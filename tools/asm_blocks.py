"""Per kernel of an AMDGPU assembly file (hipcc --cuda-device-only -S): registers, spills, scratch, and per basic block with at
least --min-mfma MFMAs the number of scratch loads / stores.    python tools/asm_blocks.py file.s [--kernels substr ...] [--min-mfma N] [--all-scratch]"""
import argparse
import re
import subprocess

ap = argparse.ArgumentParser()
ap.add_argument("asm")
ap.add_argument("--kernels", nargs="*", default=["k_sweep"])
ap.add_argument("--min-mfma", type=int, default=32)
ap.add_argument("--all-scratch", action="store_true", help="also list every block that holds scratch instructions")
args = ap.parse_args()
text = open(args.asm).read()
funcs = re.split(r"\n(?=_Z[A-Za-z0-9_]+:)", text)
heads = [(re.match(r"_Z[A-Za-z0-9_]+", f) or [""])[0] for f in funcs]
names = [h for h in heads if h]
dm = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()))
for f, h in zip(funcs, heads):
    name = dm.get(h, h)
    if not h or not any(k in name for k in args.kernels):
        continue
    body = f.split(".Lfunc_end")[0]
    meta = {}
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "sgpr_spill_count", "vgpr_spill_count", "NumSgprs"):
        m = re.search(r";\s*\.?%s:?\s*(\d+)" % key, f)
        if m:
            meta[key] = int(m.group(1))
    parts = re.split(r"\n(\.LBB[0-9_]+):", body)
    labels = ["(entry)"] + parts[1::2]
    bodies = [parts[0]] + parts[2::2]
    tot_m = len(re.findall(r"\bv_mfma_", body))
    tot_s = len(re.findall(r"\bscratch_(load|store)", body))
    calls = len(re.findall(r"\bs_swappc_b64", body))
    print("%s\n  %s\n  %d MFMAs, %d scratch instructions, %d calls, %d basic blocks" % (re.sub(r"\(.*", "", name.replace("void ", "")), meta, tot_m, tot_s, calls, len(labels)))
    hot = 0
    for lab, bb in zip(labels, bodies):
        m = len(re.findall(r"\bv_mfma_", bb))
        sl = len(re.findall(r"\bscratch_load", bb))
        ss = len(re.findall(r"\bscratch_store", bb))
        if m < args.min_mfma and not (args.all_scratch and sl + ss):
            continue
        n = len([l for l in bb.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))])
        loop = bool(re.search(r"s_cbranch_\w+\s+%s\b" % re.escape(lab), bb))
        if m >= args.min_mfma:
            hot += sl + ss
        print("    %-14s %4d MFMAs %5d instr  scratch ld %3d st %3d%s" % (lab, m, n, sl, ss, "  (loop)" if loop else ""))
    print("  scratch instructions inside blocks with >= %d MFMAs: %d of %d" % (args.min_mfma, hot, tot_s))
