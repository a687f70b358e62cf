"""VGPR pressure inside ONE basic block of an AMDGPU assembly file, from the instructions' operands (no compiler needed):
a register is counted live from its first definition in the block (or from the block's start when it is read before being written
there) to its last read in the block (to the block's end when it is never read after its last write: assumed live-out).
    python tools/asm_pressure.py file.s <mangled function> <.LBB label>"""
import re
import sys

text = open(sys.argv[1]).read()
fn, lab = sys.argv[2], sys.argv[3]
body = text[text.index("\n" + fn + ":"):]
body = body[:body.index(".Lfunc_end")]
blk = body[body.index("\n" + lab + ":"):]
m = re.search(r"\n\.LBB[0-9_]+:", blk[5:])
blk = blk[:m.start() + 5] if m else blk
ins = [l.strip() for l in blk.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def regs(tok):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out += list(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.append(int(a))
    return out


first_def, last_use, used_before_def, last_def = {}, {}, set(), {}
for k, l in enumerate(ins):
    l = l.split(";")[0]
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    stores = op.startswith(("global_store", "scratch_store", "ds_write", "flat_store", "buffer_store", "global_atomic", "ds_add", "s_", "v_cmp", "v_writelane"))
    dst = [] if stores else (regs(ops[0]) if ops else [])
    src = regs(",".join(ops if stores else ops[1:]))
    if op.startswith("v_mfma") or op.startswith("v_fma") or op.startswith("v_mac"):
        pass
    for r in src:
        if r not in first_def:
            used_before_def.add(r)
        last_use[r] = k
    for r in dst:
        first_def.setdefault(r, k)
        last_def[r] = k
n = len(ins)
live = [0] * (n + 1)
allr = set(first_def) | used_before_def
for r in allr:
    a = 0 if r in used_before_def else first_def[r]
    b = last_use.get(r, -1)
    if r in last_def and last_def[r] >= b:
        b = n  # written last: live-out
    if r in used_before_def and r not in last_def:
        b = n  # read-only here: loop-invariant, live through
    for k in range(a, min(b, n) + 1):
        live[k] += 1
print("%s %s: %d instructions, registers touched %d, live-in %d, max live %d (at instruction %d: %s)" % (fn, lab, n, len(allr), len(used_before_def), max(live), live.index(max(live)), ins[min(live.index(max(live)), n - 1)][:60]))
ro = sorted(r for r in used_before_def if r not in last_def)
print("read-only (live through) registers: %d: %s" % (len(ro), ro))
