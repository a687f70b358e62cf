"""The critical path of one SPD inverse from a GDCA_SWEEP_TRACE file (round 6: every main-list item is stamped -- taken, end of its wait, done --
beside the chain's items).  Walks back from the item that finished last: an item that had to wait was held up by the producer that finished
last before its wait ended (among the items it can depend on: same or previous update, a row or column block in common -- or the chain);
an item that did not wait was held up by its own workgroup (the item that workgroup did before).  Prints where the time of that path went.
usage: critical_path.py trace.txt"""
import sys, collections, bisect
path = sys.argv[1]
hdr = open(path).readline().split()
nblk = int(hdr[2])
main, chain = [], []
for l in open(path):
    if l.startswith("m "):
        f = l.split()
        kind = int(f[2])
        if kind == 3:
            continue
        main.append(dict(id="m%s" % f[1], kind=kind, p=int(f[3]), a=int(f[4]), b=int(f[5]), wg=int(f[6]), taken=float(f[7]), ready=float(f[8]), done=float(f[9])))
    elif not l.startswith("#"):
        q, e, a, b = l.split()
        chain.append(dict(id="c%s.%s" % (q, e), kind=100, p=int(q), a=int(e), b=0, wg=-1, taken=float(a), ready=float(a), done=float(b)))
items = main + chain
names = {0: "panel", 1: "tile", 2: "write-back", 4: "early tile", 100: "chain"}
# producers an item can depend on: previous or same update and a block index in common (chain items: any chain item, and main items of rows b0+1 .. b0+3)
by_done = sorted(items, key=lambda x: x["done"])
dones = [x["done"] for x in by_done]
def blocks(x):
    if x["kind"] == 100:
        return None
    if x["kind"] == 2:
        return None
    return {x["a"], x["b"]} if x["kind"] in (1, 4) else {x["a"]}
def may_depend(x, y):
    if y is x or y["p"] > x["p"] or y["p"] < x["p"] - 1:
        return False
    bx, by = blocks(x), blocks(y)
    if bx is None or by is None:
        return True
    if x["kind"] in (1, 4) and y["kind"] == 0:
        return y["p"] == x["p"] and y["a"] in bx
    return bool(bx & by)
def blocker(x):
    i = bisect.bisect_right(dones, x["ready"] + 0.05) - 1
    while i >= 0 and dones[i] > x["ready"] - 6.0:
        if may_depend(x, by_done[i]):
            return by_done[i]
        i -= 1
    return None
by_wg = collections.defaultdict(list)
for x in main:
    by_wg[x["wg"]].append(x)
for v in by_wg.values():
    v.sort(key=lambda x: x["taken"])
prev_of = {}
for v in by_wg.values():
    for i in range(1, len(v)):
        prev_of[v[i]["id"]] = v[i - 1]
last = max(items, key=lambda x: x["done"])
t_end = last["done"]
cur, spent, hops, steps = last, collections.Counter(), collections.Counter(), []
while cur is not None and cur["done"] > 5.0:
    waited = cur["ready"] - cur["taken"]
    spent["%s: execution" % names.get(cur["kind"], cur["kind"])] += cur["done"] - cur["ready"]
    if waited > 1.0 or cur["kind"] == 100:   # (a chain item's first stamp IS the end of its wait)
        b = blocker(cur)
        if b is None:
            spent["waits nobody accounts for"] += waited
            b = prev_of.get(cur["id"])
            if b is None and cur["kind"] == 100:   # the chain worker's previous item: the latest chain item done before this one began
                cands = [y for y in chain if y["done"] <= cur["ready"] + 0.05 and y is not cur]
                b = max(cands, key=lambda y: y["done"]) if cands else None
                if b is not None:
                    spent["chain idle between two of its items on the path"] += max(0.0, cur["ready"] - b["done"])
        else:
            spent["flag hop (producer done -> consumer sees it)"] += max(0.0, cur["ready"] - b["done"])
            hops["%s <- %s" % (names.get(cur["kind"]), names.get(b["kind"]))] += 1
        steps.append((cur, b, waited))
        cur = b
    else:
        b = prev_of.get(cur["id"])
        if b is not None:
            spent["pick-up (the item waited in the list for a workgroup)"] += max(0.0, cur["taken"] - b["done"])
            hops["%s picked up behind %s" % (names.get(cur["kind"]), names.get(b["kind"]))] += 1
        steps.append((cur, b, 0.0))
        cur = b
print("inverse %.1f us, %d blocks; the path back from the last item has %d items" % (t_end, nblk, len(steps)))
tot = sum(spent.values())
for k, v in spent.most_common():
    print("  %7.1f us  %5.1f %%  %s" % (v, 100.0 * v / t_end, k))
print("  %7.1f us accounted for" % tot)
print("links:", ", ".join("%s x%d" % kv for kv in hops.most_common(12)))
mid = [s for s in steps if 0.3 * t_end < s[0]["done"] < 0.7 * t_end][:40]
for c, b, w in reversed(mid):
    print("   %-10s %-11s p %2d (%2d,%2d) taken %8.1f ready %8.1f done %8.1f   <- %s" % (c["id"], names.get(c["kind"]), c["p"], c["a"], c["b"], c["taken"], c["ready"], c["done"],
          "%s %s p %d (%d,%d) done %.1f" % (b["id"], names.get(b["kind"]), b["p"], b["a"], b["b"], b["done"]) if b else "-"))
