"""Stage-by-stage GPU-vs-oracle report (diagnostic; prints, never asserts).  Run on the GPU box:
    python tools/gpu_check.py [quick|full]
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gaussdca.jl_amd as g  # noqa: E402
from gdca_testutil import random_msa  # noqa: E402
from oracle import gdca_oracle as o  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def stage(name, fn):
    t = time.time()
    try:
        msg = fn()
        print(f"[{name}] {msg}  ({time.time() - t:.2f}s)", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"[{name}] EXCEPTION {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()


def check_case(ctx, M, N, q=21, seed=1, pc=0.8, theta="auto"):
    print(f"\n=== case M={M} N={N} q={q} pc={pc} theta={theta}", flush=True)
    rng = np.random.default_rng(seed)
    Zo = random_msa(rng, M, N, q)  # (M, N) oracle layout
    Z = np.asfortranarray(Zo.T)    # (N, M) Julia layout
    s = q - 1
    W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, theta)
    n_o = o.neighbour_counts(Zo, thr_o)

    def s_theta():
        ps = g.pair_identity_sum(Z, ctx=ctx)
        th = g.compute_theta(Z, ctx=ctx)
        return f"pair_sum gpu={ps} oracle={o.pair_identity_sum(Zo)} theta gpu={th!r} oracle={o.compute_theta(Zo)!r}"

    stage("theta", s_theta)

    def s_ham():
        n = g.neighbour_counts(Z, thr_o, ctx=ctx)
        bad = int(np.count_nonzero(n != n_o))
        return f"thresh={thr_o} mismatches={bad}/{M} sum gpu={int(n.sum())} oracle={int(n_o.sum())} max={int(n_o.max())}"

    stage("hamming", s_ham)

    def s_w():
        W, Meff, th, thr = g.compute_weights(Z, q, theta, ctx=ctx, return_theta=True)
        return (f"W equal={np.array_equal(W, W_o)} Meff gpu={Meff!r} oracle={Meff_o!r} "
                f"theta equal={th == th_o} thresh {thr}/{thr_o}")

    stage("weights", s_w)

    Pi_o, Pij_o = o.compute_frequencies(Zo, q, W_o, Meff_o)

    def s_freq():
        Pi, Pij = g.compute_weighted_frequencies(Z, W_o, Meff_o, ctx=ctx)
        return (f"Pi rel={rel(Pi, Pi_o):.2e} Pij rel={rel(Pij, Pij_o):.2e} sym={np.array_equal(Pij, Pij.T)}")

    stage("frequencies", s_freq)

    Pi2_o, Pij2_o = o.add_pseudocount(Pi_o, Pij_o, pc, q)
    C_o = o.compute_C(Pi2_o, Pij2_o)

    def s_pc():
        Pi2, Pij2 = g.add_pseudocount(Pi_o, Pij_o, pc, q, ctx=ctx)
        Cg = g.compute_C(Pi2_o, Pij2_o, ctx=ctx)
        return (f"Pi' equal={np.array_equal(Pi2, Pi2_o)} Pij' equal={np.array_equal(Pij2, Pij2_o)} "
                f"(rel {rel(Pij2, Pij2_o):.1e}) C equal={np.array_equal(Cg, C_o)} (rel {rel(Cg, C_o):.1e})")

    stage("pseudocount+C", s_pc)

    mJ_o = o.spd_inverse(C_o)

    def s_inv():
        mJ = g.inv_cholesky(C_o, ctx=ctx)
        n = C_o.shape[0]
        res = np.linalg.norm(C_o @ mJ - np.eye(n)) / np.sqrt(n)
        return f"n={n} mJ rel={rel(mJ, mJ_o):.2e} resid={res:.2e} sym={np.array_equal(mJ, mJ.T)}"

    stage("spd_inverse", s_inv)

    def s_fn():
        S = g.compute_FN(mJ_o, q, ctx=ctx)
        return f"FN rel={rel(S, o.compute_FN(mJ_o, q)):.2e}"

    stage("FN", s_fn)

    def s_di():
        S = g.compute_DI_gauss(mJ_o, C_o, q, ctx=ctx)
        return f"DI rel={rel(S, o.compute_DI_gauss(mJ_o, C_o, q)):.2e}"

    stage("DI", s_di)

    def s_apc():
        S0 = o.compute_FN(mJ_o, q)
        return f"APC rel={rel(g.correct_APC(S0, ctx=ctx), o.correct_APC(S0)):.2e}"

    stage("APC", s_apc)

    for score, sc in (("frob", 0), ("DI", 1)):
        def s_run():
            S, st = ctx.run(Z, q, pc, -1.0 if theta == "auto" else float(theta), sc)
            S_o = o.scores_from_Z(Zo, q, pc, theta, score)
            off = ~np.eye(N, dtype=bool)
            relmax = float(np.max(np.abs(S[off] - S_o[off]) / np.maximum(np.abs(S_o[off]), 1e-3 * np.abs(S_o).max())))
            return (f"fused {score}: rel(max-norm)={rel(S, S_o):.2e} elementwise={relmax:.2e} Meff={st['Meff']!r} "
                    f"ms total={st['ms_total']:.2f} w={st['ms_weights']:.2f} cov={st['ms_covariance']:.2f} "
                    f"inv={st['ms_inverse']:.2f} (upd {st['ms_inverse_update']:.2f}) score={st['ms_score']:.2f}")

        stage("run", s_run)


def timing_case(ctx, M, N, q=21, seed=5, reps=2):
    print(f"\n=== timing M={M} N={N}", flush=True)
    rng = np.random.default_rng(seed)
    Zo = random_msa(rng, M, N, q)
    Z = np.asfortranarray(Zo.T)
    for score, sc, pc in (("frob", 0, 0.8), ("DI", 1, 0.2)):
        for r in range(reps):
            try:
                S, st = ctx.run(Z, q, pc, -1.0, sc)
                tf = st["inverse_flops"] / (st["ms_inverse"] * 1e-3) / 1e12
                tu = st["update_flops"] / max(1e-9, st["ms_inverse_update"] * 1e-3) / 1e12
                print(f"  {score} rep{r}: total={st['ms_total']:.2f}ms theta={st['ms_theta']:.2f} "
                      f"weights={st['ms_weights']:.2f} cov={st['ms_covariance']:.2f} inverse={st['ms_inverse']:.2f} "
                      f"(update {st['ms_inverse_update']:.2f}, {tu:.1f} TF in-kernel) score={st['ms_score']:.2f} | "
                      f"inverse {tf:.1f} TFLOP/s  Meff={st['Meff']:.2f} theta={st['theta']:.4f} info={st['info']} "
                      f"finite={bool(np.isfinite(S).all())}", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"  {score}: EXCEPTION {type(e).__name__}: {e}", flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    ctx = g.Context(0)
    import ctypes as C
    tf = C.c_double()
    ctx.check(ctx.lib.gdca_probe_mfma_f64(ctx.h, 20000, C.byref(tf)))
    print(f"f64 MFMA probe: {tf.value:.1f} TFLOP/s", flush=True)
    check_case(ctx, 64, 12, seed=1)
    check_case(ctx, 300, 40, seed=2, pc=0.5)
    check_case(ctx, 257, 33, q=8, seed=3, pc=0.3, theta=0.25)
    if mode != "quick":
        check_case(ctx, 1500, 100, seed=4, pc=0.8)
        timing_case(ctx, 10000, 128)
        timing_case(ctx, 50000, 500)


if __name__ == "__main__":
    main()
