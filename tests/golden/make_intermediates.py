"""Regenerates tests/golden/intermediates.json from the oracle (see README.md)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gdca_oracle as o  # noqa: E402

CASES = {
    "small.FNRout.txt": dict(fasta="small.fasta.gz", kw={}),
    "small.DIRout.txt": dict(fasta="small.fasta.gz", kw=dict(pseudocount=0.2, score="DI", remove_dups=True)),
    "small.DIRout2.txt": dict(fasta="small.fasta.gz",
                              kw=dict(pseudocount=0.2, score="DI", theta=0.0, max_gap_fraction=0.8,
                                      min_separation=4)),
    "large.DIRout.txt": dict(fasta="large.fasta.gz", kw=dict(pseudocount=0.2, score="DI", remove_dups=True)),
}

out = {}
for golden, c in CASES.items():
    kw = c["kw"]
    Z = o.read_fasta_alignment(os.path.join(HERE, "reference", c["fasta"]), kw.get("max_gap_fraction", 0.9))
    if kw.get("remove_dups"):
        Z, _ = o.remove_duplicate_sequences(Z)
    W, Meff, th, thresh = o.compute_weights(Z, kw.get("theta", "auto"))
    n = o.neighbour_counts(Z, thresh)
    out[golden] = dict(fasta=c["fasta"], kwargs=kw, N=int(Z.shape[1]), M=int(Z.shape[0]), q=int(Z.max()),
                       theta=th, thresh=thresh, Meff=Meff, pair_identity_sum=o.pair_identity_sum(Z),
                       neighbour_count_sum=int(n.sum()), neighbour_count_max=int(n.max()),
                       # sum(W) three ways (oracle.meff_three_ways): which one DCAUtils' pairwise SIMD sum returns only Julia can say
                       Meff_candidates=o.meff_three_ways(W))
with open(os.path.join(HERE, "intermediates.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
