"""Register / scratch accounting of every kernel of libgdca.so, from the compiler (no GPU needed):

  * `-Rpass-analysis=kernel-resource-usage`: VGPRs, AGPRs, spills, scratch bytes per lane, occupancy, LDS per kernel;
  * `-S`: per basic block of the chosen kernels, how many MFMAs and how many scratch_load / scratch_store instructions it holds --
    the question VERDICT r03 #3 asked of `k_sweep<true>`: does spill traffic sit INSIDE the MFMA loops or around them?

    python tools/kernel_resources.py [--kernels k_sweep k_sweep_merged] [--min-mfma 32] > profiles/rNN_kernel_resources.txt
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussdca.jl_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function", "-Wno-pass-failed",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

ap = argparse.ArgumentParser()
ap.add_argument("--sources", nargs="*", default=["k_inverse.hip", "k_hamming.hip", "k_hamming_fp4.hip", "k_tally.hip", "k_score.hip", "k_theta.hip", "k_elementwise.hip", "k_rank.hip"])
ap.add_argument("--kernels", nargs="*", default=["k_sweep", "k_sweep_merged"], help="kernels whose basic blocks are listed (substring of the demangled name)")
ap.add_argument("--min-mfma", type=int, default=32, help="list basic blocks with at least this many MFMAs")
args = ap.parse_args()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


print("# kernel resource usage (hipcc %s)\n" % " ".join(FLAGS[:5]))
print("%-78s %5s %5s %7s %7s %8s %5s %8s" % ("kernel", "VGPR", "AGPR", "VGPRsp", "SGPRsp", "scratchB", "occ", "LDS B"))
with tempfile.TemporaryDirectory() as tmp:
    for src in args.sources:
        path = os.path.join(CSRC, src)
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.path.join(tmp, "x.o")],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(src, "does not compile:", r.stderr[-500:])
            sys.exit(1)
        blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
        names = [b.split()[0] for b in blocks]
        dm = demangle(names)
        for b, nm in zip(blocks, names):
            def field(label):
                m = re.search(label + r": (\d+)", b)
                return int(m.group(1)) if m else -1
            shown = dm.get(nm, nm).replace("void ", "").replace("(anonymous namespace)::", "")
            shown = re.sub(r"\(.*", "", shown)
            print("%-78s %5d %5d %7d %7d %8d %5d %8d" % ((src + ": " + shown)[:78], field("VGPRs"), field("AGPRs"), field("VGPRs Spill"), field("SGPRs Spill"),
                                                         field(r"ScratchSize \[bytes/lane\]"), field(r"Occupancy \[waves/SIMD\]"), field(r"LDS Size \[bytes/block\]")))

    # ---- basic blocks of the sweep kernels ----
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", os.path.join(CSRC, "k_inverse.hip"), "-o", os.path.join(tmp, "k.s")],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print("-S failed:", r.stderr[-500:])
        sys.exit(1)
    text = open(os.path.join(tmp, "k.s")).read()
funcs = re.split(r"\n(?=_Z[A-Za-z0-9_]+:)", text)
heads = [(re.match(r"_Z[A-Za-z0-9_]+", f) or [""])[0] for f in funcs]   # (the first piece is the file's preamble: no name)
dm = demangle([h for h in heads if h])
print("\n# basic blocks of the sweep kernels with >= %d MFMAs: where the scratch instructions are\n" % args.min_mfma)
for f, h in zip(funcs, heads):
    name = dm.get(h, h).replace("(anonymous namespace)::", "")
    if not any(k in name for k in args.kernels):
        continue
    if not re.search(r"^\s+s_endpgm", f, re.M):
        continue
    body = f.split(".Lfunc_end")[0]
    parts = re.split(r"\n(\.LBB[0-9_]+):", body)
    tot_m = len(re.findall(r"\bv_mfma_", body))
    tot_s = len(re.findall(r"\bscratch_(load|store)", body))
    name = name.replace("void ", "")
    print("%s\n  whole kernel: %d MFMAs, %d scratch instructions, %d basic blocks" % (re.sub(r"\(.*", "", name), tot_m, tot_s, len(parts) // 2 + 1))
    labels = ["(entry)"] + parts[1::2]
    bodies = [parts[0]] + parts[2::2]
    in_hot = 0
    for lab, bb in zip(labels, bodies):
        m = len(re.findall(r"\bv_mfma_", bb))
        if m < args.min_mfma:
            continue
        sl = len(re.findall(r"\bscratch_load", bb))
        ss = len(re.findall(r"\bscratch_store", bb))
        n = len([l for l in bb.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))])
        loop = bool(re.search(r"s_cbranch_\w+\s+%s\b" % re.escape(lab), bb))
        in_hot += sl + ss
        print("    %-14s %4d MFMAs %5d instructions  scratch loads %3d stores %3d%s" % (lab, m, n, sl, ss, "   <- branches back to itself (a loop)" if loop else ""))
    print("  scratch instructions inside those blocks: %d of %d" % (in_hot, tot_s))
