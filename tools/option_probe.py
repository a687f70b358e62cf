"""Inverse-stage time over alignment lengths N for combinations of context options, ONE process and context (the options of a
context take effect at its next call), alternating the combinations inside every repetition so that clock drift hits them alike.
usage: option_probe.py N,N,... "KEY=V,KEY=V;KEY=V;..." [reps [stat [M]]]   (an empty combination = the defaults)
Prints min and median ms_inverse per (N, combination) and whether the scores are bit-identical with the first combination's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth

Ns = [int(x) for x in sys.argv[1].split(",")]
combos = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[2].split(";")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
stat = sys.argv[4] if len(sys.argv) > 4 else "ms_inverse"     # the stage time to report (ms_weights, ms_covariance, ms_total ...)
Mseq = int(sys.argv[5]) if len(sys.argv) > 5 else 3000        # sequences per family
DEFAULTS = {"GROUP": "-1", "MCUS": "-1", "REM_TAIL": "-1", "PANEL_HALVES": "-1", "RAMP": "1", "RING": "8", "RAGGED": "1", "MCU_SOLO": "-1",
            "TALLY_TJ": "0", "HAMMING_MODE": "auto"}
ctx = g.Context(0)
print(stat + " min / median; combos: " + " | ".join(",".join("%s=%s" % kv for kv in c.items()) or "default" for c in combos))
for N in Ns:
    Z = np.asfortranarray(synth.synth_family(N, Mseq, 21, 7 + N).T)
    t = [[] for _ in combos]
    ref, same = None, []
    for rep in range(reps + 1):
        for ci, c in enumerate(combos):
            for k, v in {**DEFAULTS, **c}.items():
                ctx.set_option(k, v)
            S, st = ctx.run(Z, 21, 0.8, 0.3, 0)
            if rep == 0:   # warm-up round: compare the results
                if ref is None:
                    ref = S.copy()
                same.append(bool(np.array_equal(S, ref)) or float(np.max(np.abs(S - ref))))
            else:
                t[ci].append(st[stat])
    nblk = -(-N * 20 // 128)
    print("N %4d nblk %3d " % (N, nblk) + "  ".join("%7.3f/%7.3f" % (min(x), float(np.median(x))) for x in t) + "   same: %s" % same, flush=True)
