"""Loop structure of an AMDGPU assembly file (hipcc --cuda-device-only -S) from the compiler's own loop annotations
("=>This Loop Header", "Parent Loop", "in Loop: Header="): per function, the loops that hold an `s_barrier` and whether any of them is
DIVERGENT -- a loop whose latch narrows the exec mask (`s_andn2_b64 exec, exec, <lanes that left>`) so that lanes leave it one by one.

Why: a workgroup barrier inside a loop that lanes can leave separately is malformed by construction -- the lanes that stay keep
arriving at barriers the others never reach again, or (what round 4 saw) thread 0's duties of an iteration are skipped.  hipcc builds
such a loop out of source that LOOKS uniform when a persistent loop has two thread-0-only blocks in front of one back edge: the
per-lane continue mask is then made from the `tid == 0` compare (the `-DGDCA_EXP_LATE_NEXT` form of k_sweep_merged:
wrong inverses in round 4, a hang on an empty list in round 5).  tests/test_kernel_resources.py asserts that no function of
k_inverse.hip holds such a loop and that the late form trips this very check.

    python tools/asm_loops.py file.s [--all]"""
import re
import sys


def functions(text):
    for f in re.split(r"\n(?=_Z[A-Za-z0-9_]+:)", text):
        m = re.match(r"(_Z[A-Za-z0-9_]+):", f)
        if m:
            yield m.group(1), f.split(".Lfunc_end")[0]


def loops_of(body):
    """{loop header: {"parent", "depth", "barriers", "mfmas", "narrowing": [labels of this loop's own blocks that narrow exec]}}"""
    parts = re.split(r"\n(\.LBB[0-9_]+):", body)
    labels, bodies = ["entry"] + parts[1::2], [parts[0]] + parts[2::2]
    loops, inner = {}, {}
    for lab, b in zip(labels, bodies):
        pre = []
        for ln in b.split("\n"):
            if ln.strip().startswith(";") or not ln.strip():
                pre.append(ln)
            else:
                break
        head = "\n".join(pre)
        key = lab.replace(".L", "")
        m = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", head)
        if m:
            par = re.findall(r"Parent Loop (BB[0-9_]+) Depth=(\d+)", head)
            loops[key] = {"parent": max(par, key=lambda x: int(x[1]))[0] if par else None, "depth": int(m.group(1)), "barriers": 0, "mfmas": 0,
                          "narrowing": []}
            inner[lab] = key
        else:
            m = re.search(r"in Loop: Header=(BB[0-9_]+) Depth=\d+", head)
            if m:
                inner[lab] = m.group(1)
    for lab, b in zip(labels, bodies):
        h = inner.get(lab)
        if h not in loops:
            continue
        # a labelled section may run on past a branch: what follows a (conditional) back edge to the header of loop L without a label of
        # its own is the fall-through EXIT of L -- it belongs to L's parent at most (k_rank_scatter: the barrier behind a per-lane
        # zeroing loop sits in the same labelled section as the loop's single block)
        segs = re.split(r"^(\s+s_c?branch\w*\s+\.LBB[0-9_]+)[^\n]*$", b, flags=re.M)
        cur = h
        for k in range(0, len(segs), 2):
            seg = segs[k]
            nb, nm = len(re.findall(r"^\s+s_barrier", seg, re.M)), len(re.findall(r"\bv_mfma_", seg))
            a = cur
            while a is not None and a in loops:
                loops[a]["barriers"] += nb
                loops[a]["mfmas"] += nm
                a = loops[a]["parent"]
            if cur in loops and re.search(r"^\s+s_andn2_b64 exec, exec,", seg, re.M):
                loops[cur]["narrowing"].append(lab)
            if k + 1 < len(segs) and cur in loops:
                tgt = re.search(r"\.L(BB[0-9_]+)", segs[k + 1]).group(1)
                a = cur
                while a is not None and a in loops:
                    if a == tgt:
                        cur = loops[a]["parent"]
                        break
                    a = loops[a]["parent"]
    return loops


def divergent_barrier_loops(text):
    """[(function, loop header, depth, barriers, mfmas, narrowing blocks)] over every function of the file"""
    bad, seen = [], 0
    for name, body in functions(text):
        for h, v in loops_of(body).items():
            seen += 1
            if v["barriers"] and v["narrowing"]:
                bad.append((name, h, v["depth"], v["barriers"], v["mfmas"], v["narrowing"]))
    return bad, seen


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    for name, body in functions(text):
        L = loops_of(body)
        wb = {h: v for h, v in L.items() if v["barriers"]}
        if not wb and "--all" not in sys.argv:
            continue
        print("%s: %d loops, %d hold a barrier" % (name[:70], len(L), len(wb)))
        for h, v in wb.items():
            print("    %-12s depth %d  barriers %3d  MFMAs %5d  %s" % (h, v["depth"], v["barriers"], v["mfmas"],
                                                                      "DIVERGENT: exec narrowed in " + ", ".join(v["narrowing"]) if v["narrowing"] else "uniform"))
    bad, seen = divergent_barrier_loops(text)
    print("%d loops, %d divergent loops that hold a barrier" % (seen, len(bad)))
    sys.exit(1 if bad else 0)
