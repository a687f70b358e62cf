import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussdca.jl_amd as g
np.set_printoptions(linewidth=220, precision=2)
ctx = g.Context(0)
rng = np.random.default_rng(1)
def spd(n):
    B = rng.standard_normal((n, 2 * n))
    return B @ B.T / (2 * n) + 0.3 * np.eye(n)
A0, A1 = spd(128), spd(128)
for eps in (0.0, 1e-6, 1e-2):
    A = np.zeros((256, 256))
    A[:128, :128] = A0
    A[128:, 128:] = A1
    Cc = rng.standard_normal((128, 128)) * eps
    A[128:, :128] = Cc
    A[:128, 128:] = Cc.T
    try:
        X = g.inv_cholesky(A, ctx=ctx)
        Xr = np.linalg.inv(A)
        E = np.abs(X - Xr) / np.abs(Xr).max()
        print("eps", eps, "ok, max err %.2e" % E.max(), "block errs", E.reshape(2, 128, 2, 128).max(axis=(1, 3)))
    except Exception as e:
        print("eps", eps, "EXC", e)
