"""gaussdca.jl_amd -- MI355X (gfx950) Gaussian-DCA hot path behind GaussDCA.jl's interface.

``gDCA`` / ``printrank`` mirror the reference module's exports (src/GaussDCA.jl:3); the
DCAUtils-named operators are in :mod:`dcautils`.  All hot-path arithmetic runs in the HIP
library ``libgdca.so`` (C-ABI: include/gdca.h); importing this package never falls back to a
CPU implementation -- calling any operator without the built library or without a GPU raises.
"""
from ._lib import run_dev_phased, run_ranked_phased_async, spd_inverse_batch_dev  # noqa: F401
from ._lib import (ArgumentError, Context, ConvergenceError, DeviceBuffer, GdcaError, PosDefException,  # noqa: F401
                   default_context, load)
from .dcautils import (add_pseudocount, compute_C, compute_DI_gauss, compute_FN, compute_ranking,  # noqa: F401
                       compute_theta, compute_weighted_frequencies, compute_weights, correct_APC,
                       inv_cholesky, neighbour_counts, pair_identity_sum, printrank,
                       read_fasta_alignment, remove_duplicate_sequences, Ranking)
from .gdca import check_arguments, gDCA  # noqa: F401

__all__ = ["gDCA", "printrank"]
