"""Debug aid for the pivot of the sweep kernel: the SPD inverse of one 128 x 128 block (a one-block inverse) against numpy, with the
error reported per 16 x 16 micro-tile."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussdca.jl_amd as g

np.set_printoptions(linewidth=200, precision=2, suppress=False)
ctx = g.Context(0)
rng = np.random.default_rng(0)
for name in ("diag", "blockdiag16", "spd"):
    n = 128
    if name == "diag":
        A = np.diag(1.0 + np.arange(n) / 16.0)
    elif name == "blockdiag16":
        A = np.zeros((n, n))
        for b in range(8):
            B = rng.standard_normal((16, 40))
            A[16 * b:16 * b + 16, 16 * b:16 * b + 16] = B @ B.T / 40 + 0.5 * np.eye(16)
    else:
        B = rng.standard_normal((n, 300))
        A = B @ B.T / 300 + 0.3 * np.eye(n)
    try:
        X = g.inv_cholesky(A, ctx=ctx)
    except Exception as e:  # noqa: BLE001
        print(name, "EXC", e)
        continue
    Xr = np.linalg.inv(A)
    E = np.abs(X - Xr) / np.max(np.abs(Xr))
    T = E.reshape(8, 16, 8, 16).max(axis=(1, 3))
    print(name, "max rel err %.3e  sym %s" % (E.max(), np.array_equal(X, X.T)))
    print(np.where(T > 1e-10, T, 0.0))
    if name == "blockdiag16" and E.max() > 1e-10:
        b = 0
        print("tile(0,0) got\n", X[:6, :6], "\nwant\n", Xr[:6, :6])
