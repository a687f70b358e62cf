import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussdca.jl_amd as g
from oracle import gdca_oracle as o

np.set_printoptions(linewidth=220, precision=1)
ctx = g.Context(0)
rng = np.random.default_rng(1)
mats = {}
for n in (256, 384, 1152):
    B = rng.standard_normal((n, 2 * n))
    mats["rand%d" % n] = B @ B.T / (2 * n) + 0.3 * np.eye(n)
Z = o.read_fasta_alignment("tests/golden/reference/small.fasta.gz", 0.9)
Pi, Pij, Meff, W = o.compute_weighted_frequencies(Z, 21, "auto")
C = o.compute_C(*o.add_pseudocount(Pi, Pij, 0.8, 21))
mats["small_cov"] = C
mats["small_cov_block0"] = C[:128, :128].copy()
mats["small_cov_256"] = C[:256, :256].copy()
for name, A in mats.items():
    n = A.shape[0]
    try:
        X = g.inv_cholesky(A, ctx=ctx)
    except Exception as e:  # noqa: BLE001
        print(name, "EXC", e)
        continue
    Xr = np.linalg.inv(A)
    E = np.abs(X - Xr) / np.max(np.abs(Xr))
    nb = (n + 127) // 128
    Ep = np.zeros((nb * 128, nb * 128))
    Ep[:n, :n] = E
    T = Ep.reshape(nb, 128, nb, 128).max(axis=(1, 3))
    print(name, "n", n, "cond %.1e" % np.linalg.cond(A), "max rel err %.3e" % E.max())
    if E.max() > 1e-9:
        print(T)
