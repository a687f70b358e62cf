"""Per-step timeline of the chain between single blocks from a GDCA_SWEEP_TRACE file: when each kind of item of step q became
ready (the end of its dependency wait) and ended, and the main-list items of row block b0 + 3 that the chain waits for two steps later.
usage: chain_timeline.py trace.txt [first_step last_step]"""
import sys
path = sys.argv[1]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 14)
hdr = open(path).readline()
new_order = "xslab(q-1)" in hdr
items, xs = {}, {}
for l in open(path):
    if l.startswith("# x "):
        p = int(l.split(":")[0].split()[2])
        import re
        f = re.findall(r"(-?[\d.]+) \(ready\s+(-?[\d.]+)\) \.\.\s+(-?[\d.]+)", l)
        xs[p] = [tuple(float(v) for v in t) for t in f]
    elif not l.startswith("#") and not l.startswith("m "):
        q, e, a, b = l.split()
        items[(int(q), int(e))] = (float(a), float(b))
ng = max(q for q, _ in items) + 1
def grp(q, kind):   # kind: 'slab', 'xslab', 'slab2' of step q -> list of (ready, end)
    if new_order:
        if kind == 'slab':
            base = 17 if (q >= 1 and q + 1 < ng) else 1
            return [items.get((q, base + i)) for i in range(8)]
        base = 1 if kind == 'xslab' else 9
        return [items.get((q + 1, base + i)) for i in range(8)]
    base = {'slab': 1, 'xslab': 9, 'slab2': 17, 'slab3': 25}[kind]
    return [items.get((q, base + i)) for i in range(8)]
def span(v):
    v = [x for x in v if x]
    return (min(a for a, _ in v), max(b for _, b in v)) if v else (float('nan'),) * 2
print("step | pivot ready .. end | slab ready .. end | xslab ready .. end | slab2 ready .. end || main list, row b0+3 of update q: panels ready .. done | (b0+3,b0+1) ready .. done | (b0+3,b0+2) | (b0+3,b0+3)")
prev = None
for q in range(lo, min(hi, ng - 1) + 1):
    pv = items[(q, 0)]
    s = "%3d | %7.1f .. %7.1f" % (q, pv[0], pv[1])
    for k in ('slab', 'xslab', 'slab2') + (('slab3',) if (q, 25) in items else ()):
        a, b = span(grp(q, k))
        s += " | %7.1f .. %7.1f" % (a, b)
    s += " ||"
    for t in xs.get(q, []):
        s += " %7.1f .. %7.1f |" % (t[1], t[2])
    s += "   period %5.1f" % (pv[0] - prev) if prev is not None else ""
    prev = pv[0]
    print(s)
