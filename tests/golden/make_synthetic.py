"""Generates tests/golden/synthetic_*.npz: small seeded alignments with the ORACLE's outputs
(oracle/gdca_oracle.py).  They pin the oracle against accidental edits (CPU test) and let the GPU
parity tests compare against stored vectors.  Run from the repo root: python tests/golden/make_synthetic.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdca_testutil import random_msa  # noqa: E402
from oracle import gdca_oracle as o  # noqa: E402

CASES = [  # name, seed, M, N, q, theta, pseudocount
    ("synthetic_a", 101, 240, 28, 21, "auto", 0.8),
    ("synthetic_b", 202, 500, 45, 21, 0.3, 0.2),
    ("synthetic_c", 303, 130, 17, 6, "auto", 0.5),
]

for name, seed, M, N, q, theta, pc in CASES:
    rng = np.random.default_rng(seed)
    Z = random_msa(rng, M, N, q)
    Z[0, 0] = q
    W, Meff, th, thresh = o.compute_weights(Z, theta)
    n_k = o.neighbour_counts(Z, thresh)
    S_fn = o.scores_from_Z(Z, q, pc, theta, "frob")
    S_di = o.scores_from_Z(Z, q, pc, theta, "DI")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), Z=Z, q=q, theta_in=str(theta), pseudocount=pc,
                        theta=th, thresh=thresh, Meff=Meff, n_k=n_k, pair_identity_sum=o.pair_identity_sum(Z),
                        S_frob=S_fn, S_DI=S_di)
    print(name, Z.shape, "theta=%r thresh=%d Meff=%r" % (th, thresh, Meff))
