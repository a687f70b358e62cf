"""Refit the LPT cost model of gaussdca.jl_amd/batch.py (seconds: A3 n^3 + B1 b + B2 b^2 + H2 M^2 N + H1 M + T2 N^2 M + T1 N^2 + T0) to measured per-family
stage times (bench.py --config E --dump-families FILE: one family after the other) and say how well the model's makespans agree with the measured ones.
    python tools/fit_batch_model.py profiles/r06_E_per_family.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussdca.jl_amd import batch

fams = json.load(open(sys.argv[1]))["families"]
N = np.array([f["N"] for f in fams], float)
M = np.array([f["M"] for f in fams], float)
n = np.array([f["n"] for f in fams], float)
inv = np.array([f["ms_inverse"] for f in fams]) * 1e-3
ham = np.array([f["ms_weights"] + f["ms_theta"] for f in fams]) * 1e-3
cov = np.array([f["ms_covariance"] + f["ms_score"] for f in fams]) * 1e-3
tot = np.array([f["ms_total"] for f in fams]) * 1e-3
blocks = np.ceil(n / 128)
ci = np.linalg.lstsq(np.c_[n ** 3, blocks, blocks * blocks], inv, rcond=None)[0]
ch = np.linalg.lstsq(np.c_[M * M * N, M], ham, rcond=None)[0]
cc = np.linalg.lstsq(np.c_[N * N * M, N * N, np.ones_like(N)], cov, rcond=None)[0]
print("fit: A3 %.4g B1 %.4g B2 %.4g | H2 %.4g H1 %.4g | T2 %.4g T1 %.4g T0 %.4g" % (*ci, *ch, *cc))
print("batch.py: A3 %.4g B1 %.4g B2 %.4g | H2 %.4g H1 %.4g | T2 %.4g T1 %.4g T0 %.4g" % (batch.A3, batch.B1, batch.B2, batch.H2, batch.H1, batch.T2, batch.T1, batch.T0))
model = np.array([batch.family_cost(int(a), int(b)) for a, b in zip(N, M)])
print("batch.py's model against the measured per-family totals: sum %.3f s against %.3f s; per family ratio model / measured: median %.3f, 5 %% .. 95 %%: %.3f .. %.3f"
      % (model.sum(), tot.sum(), np.median(model / tot), *np.percentile(model / tot, [5, 95])))
sizes = [(int(a), int(b)) for a, b in zip(N, M)]
for world in (2, 4, 8):
    sh = batch.shard_families(sizes, world)
    pred = max(sum(model[f] for f in s) for s in sh)
    meas = max(sum(tot[f] for f in s) for s in sh)
    ideal = tot.sum() / world
    print("world %d: makespan predicted %.3f s, from the measured times %.3f s (%.1f %%); the measured makespan over the ideal %.4f" % (world, pred, meas, 100 * (pred / meas - 1), meas / ideal))
