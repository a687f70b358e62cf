"""The synthetic-family generator behind bench.py: native == numpy statement, the
regime invariants the survey lists, and FASTA round trips through the native reader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gaussdca.jl_amd import dcautils, synth  # noqa: E402
from oracle import gdca_oracle as o  # noqa: E402
import host_mirrors as hm  # noqa: E402


@pytest.mark.parametrize("N,M,q,seed", [(53, 100, 21, 1), (128, 3000, 21, 0xB128), (37, 999, 5, 7), (10, 26, 21, 0),
                                        (1, 1, 2, 3), (300, 1500, 21, 2**63 + 5)])
def test_native_matches_numpy_statement(N, M, q, seed):
    a = synth.synth_family(N, M, q, seed)
    b = hm.synth_family_py(N, M, q, seed)
    assert a.dtype == np.int8 and a.shape == (M, N)
    assert np.array_equal(a, b)
    assert a.min() >= 1 and a.max() <= q
    assert np.array_equal(a, synth.synth_family(N, M, q, seed))           # deterministic
    if M > 50 and N > 20:
        assert not np.array_equal(a, synth.synth_family(N, M, q, seed + 1))


def test_bad_arguments():
    for args in [(0, 10, 21, 1), (10, 0, 21, 1), (10, 10, 1, 1), (10, 10, 32, 1)]:
        with pytest.raises(synth._lib.ArgumentError):
            synth.synth_family(*args)


@pytest.mark.parametrize("N,M", [(128, 3000), (200, 2500)])
def test_regime_invariants(N, M):
    """>=1 gap => q=21; gap fraction < 0.9; 0.25 <= phi <= 0.45 (auto theta in 0.27-0.49);
    0.1 <= Meff/M <= 0.8; C positive definite at pc in {0.2, 0.8}."""
    Z = synth.synth_family(N, M, 21, 0xB128)
    assert Z.max() == 21
    assert (Z == 21).mean(axis=1).max() < 0.9
    theta = o.compute_theta(Z)
    assert 0.27 <= theta <= 0.49
    assert 0.25 <= 0.1216 / theta <= 0.45
    Pi, Pij, Meff, _ = o.compute_weighted_frequencies(Z, 21, theta)
    assert 0.1 <= Meff / M <= 0.8
    for pc in (0.2, 0.8):
        Pi_pc, Pij_pc = o.add_pseudocount(Pi, Pij, pc, 21)
        np.linalg.cholesky(o.compute_C(Pi_pc, Pij_pc))                      # raises if not SPD


@pytest.mark.parametrize("suffix", [".fasta", ".fasta.gz"])
def test_fasta_round_trip(tmp_path, suffix):
    Z = synth.synth_family(61, 240, 21, 99)
    path = str(tmp_path / ("fam" + suffix))
    synth.write_fasta(path, Z)
    for reader in (dcautils.read_fasta_alignment, hm.read_fasta_alignment_py, ):
        back = reader(path, 1.0)                                           # (N, M) Fortran order
        assert back.shape == (61, 240)
        assert np.array_equal(np.ascontiguousarray(back.T), Z)
    assert np.array_equal(o.read_fasta_alignment(path, 1.0), Z)
    with pytest.raises(synth._lib.ArgumentError):
        synth.write_fasta(path, np.zeros((3, 4), dtype=np.int8))           # symbol 0 has no letter
