#!/bin/bash
out=gpurun_out/r6j; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
GDCA_HAMMING_MODE=mfma rocprofv3 --kernel-trace -d $out/C_mfma -o c -- python3 bench.py --config C --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $out/C_mfma.json 2> $out/C_mfma.err
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import torch
import gaussdca.jl_amd as g
from gaussdca.jl_amd import synth
Z = synth.synth_family(500, 50000, 21, 0xC500)
# candidate densities of the two bounds on a sample of pairs
rng = np.random.default_rng(0)
k = rng.integers(0, 50000, 400000); l = rng.integers(0, 50000, 400000)
d = (Z[k] != Z[l]).sum(1)
d3 = (((Z[k] ^ Z[l]) & 7) != 0).sum(1)
x = (Z[k] ^ Z[l]) & 7
D = ((x & 1) + ((x >> 1) & 1) + ((x >> 2) & 1)).sum(1)
ctx = g.Context(0)
W, Meff, theta, thr = g.compute_weights(np.asfortranarray(Z.T), 21, 'auto', ctx=ctx) if hasattr(g, 'compute_weights') else (None, None, None, None)
print('theta', theta, 'thresh', thr)
thr = int(thr)
m = k != l
print('pairs sampled', m.sum(), 'true neighbours %.2e' % ((d[m] < thr).mean()), 'd3 candidates %.2e' % ((d3[m] < thr).mean()), 'D candidates %.2e' % ((D[m] < 3 * thr).mean()))
print('unrelated: mean d %.1f d3 %.1f D %.1f; 3 thr = %d' % (d[m].mean(), d3[m].mean(), D[m].mean(), 3 * thr))
import collections
print('D percentiles', np.percentile(D[m], [0.01, 0.1, 1, 5, 50]))
PY
