"""Property tests on random small alignments (hypothesis): GPU == oracle for every draw.
Integers bit-exact; scores within 1e-6 relative (+1e-9 of the largest score for zero crossings)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from gdca_testutil import random_msa, score_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import gaussdca.jl_amd as g
    from oracle import gdca_oracle as o

    ctx = g.Context(0)
    yield g, o, ctx
    ctx.close()


@settings(max_examples=250, derandomize=True, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(2, 300), N=st.integers(3, 40),  # N = 2: APC cancels the only score to 0/0
       q=st.sampled_from([3, 5, 21, 24]), theta=st.one_of(st.just("auto"), st.floats(0.0, 0.6)),
       pc=st.floats(0.05, 1.0), score=st.sampled_from(["frob", "DI"]))
def test_fused_path_matches_oracle(env, seed, M, N, q, theta, pc, score):
    g, o, ctx = env
    rng = np.random.default_rng(seed)
    Zo = random_msa(rng, M, N, q)
    Zo[0, 0] = q
    Z = np.asfortranarray(Zo.T)
    W_o, Meff_o, th_o, thr_o = o.compute_weights(Zo, theta)
    n_o = o.neighbour_counts(Zo, thr_o)
    assert np.array_equal(g.neighbour_counts(Z, thr_o, ctx=ctx), n_o)
    try:
        S_o = o.scores_from_Z(Zo, q, pc, theta, score)
    except o.NotPositiveDefinite:
        with pytest.raises(g.PosDefException):
            ctx.run(Z, q, pc, -1.0 if theta == "auto" else float(theta), 1 if score == "DI" else 0)
        return
    # The bar scales with the conditioning of the covariance instead of giving up on it: both inverses carry ~cond * 2^-53 of
    # relative error, so the scores may differ by a multiple of that (1e-6 up to cond ~ 1e8).  Beyond cond ~ 1e13 neither side
    # has a correct digit left: only "finite where the oracle is finite" is asked for.
    cond = np.linalg.cond(o.compute_C(*o.add_pseudocount(*o.compute_frequencies(Zo, q, W_o, Meff_o), pc, q)))
    S, stt = ctx.run(Z, q, pc, -1.0 if theta == "auto" else float(theta), 1 if score == "DI" else 0)
    assert stt["thresh"] == thr_o and stt["Meff"] == Meff_o and stt["theta"] == th_o
    if cond > 1e13:
        assert np.array_equal(np.isfinite(S), np.isfinite(S_o)) or not np.isfinite(S_o).all()
        return
    slack = max(1.0, 64.0 * cond * 2.0 ** -53 / 1e-6)
    if not np.isfinite(S_o).all():
        # degenerate input (e.g. pseudocount 1: every score is 0 and APC divides 0 by 0, in the reference too):
        # the GPU path must be non-finite in the same places
        assert np.array_equal(np.isfinite(S), np.isfinite(S_o))
        return
    if np.max(np.abs(S_o)) < 1e-10:
        # no signal at all (e.g. two all-gap sequences): the exact scores are 0 and each side holds either rounding noise of
        # the O(1) inverse-covariance entries they were computed from, or -- when that noise happens to cancel to an exact 0
        # in every pair -- the 0/0 of correct_APC (src/GaussDCA.jl:78-86 divides by the sum of all scores), as the reference
        # itself would
        nan = np.isnan(S)
        # NaN only as that all-or-nothing 0/0 of the APC denominator (S - (Sj * Si) / 0: every entry at once, the diagonal
        # included, src/GaussDCA.jl:84), never as isolated entries -- a NaN produced inside FN / DI on a degenerate input must
        # not hide behind this branch
        assert (not nan.any()) or nan.all()
        assert np.max(np.abs(np.nan_to_num(S, nan=0.0))) < 1e-10
        return
    # DI = s/2 log(1/2) + 1/2 sum_k log(1 + sqrt(1 + 4 gamma_k)) is a difference of O(s) quantities: with a pseudocount
    # near 1 the couplings vanish, the scores drop to ~1e-10 and BOTH implementations (and DCAUtils, which evaluates the
    # same expression) carry s * 2^-53-sized rounding noise in them; that floor is not a parity error
    atol_abs = 4.0 * (q - 1) * 2.0 ** -53 * 16 if score == "DI" else 0.0
    ok, max_rel, max_abs = score_close(S, S_o, rtol=1e-6 * slack, atol_frac=1e-9 * slack, atol_abs=atol_abs)
    assert ok, (max_rel, max_abs, cond)
