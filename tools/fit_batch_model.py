"""Refit the LPT cost model of gaussdca.jl_amd/batch.py (seconds: max(ALPHA n^3, CHAIN blocks) + BETA M^2 N + GAMMA N^2 M) to measured per-family
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
big = n >= 58 * 128
alpha = float(np.sum(inv[big] * n[big] ** 3) / np.sum(n[big] ** 6))
blocks = np.ceil(n / 128)
small = blocks <= 30
chain = float(np.sum(inv[small] * blocks[small]) / np.sum(blocks[small] ** 2))
beta = float(np.sum(ham * M * M * N) / np.sum((M * M * N) ** 2))
gamma = float(np.sum(cov * N * N * M) / np.sum((N * N * M) ** 2))
print("fit: ALPHA %.3g  BETA %.3g  GAMMA %.3g  CHAIN %.3g   (batch.py: %.3g %.3g %.3g %.3g)" % (alpha, beta, gamma, chain, batch.ALPHA, batch.BETA, batch.GAMMA, batch.CHAIN))
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
