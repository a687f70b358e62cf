#!/bin/bash
# A/B of the two symbol loops of gdca_inflate.cpp on the host of the GPU box (no GPU work): bash tools/experiments/inflate/ab.sh
cd $(dirname $0)/../../..
D=/tmp/gdca_inflate_ab; rm -rf $D; mkdir -p $D
gaussdca.jl_amd/gdca_cli --synth 350 42000 57360 $D/fam.fasta.gz
gaussdca.jl_amd/gdca_cli --synth 500 50000 50432 $D/famC.fasta.gz
cp gaussdca.jl_amd/csrc/gdca_inflate.cpp $D/inflate_branchy.cpp                 # the product: a branch on "literal or match"
cp tools/experiments/inflate/inflate_unified.cpp.txt $D/inflate_unified.cpp      # archived: literals and matches on one branch-free path
cp tools/experiments/inflate/inflate_twoshift.cpp.txt $D/inflate_twoshift.cpp    # archived: the product before its entries took code + extra bits in one shift
for v in branchy twoshift unified; do
  for fl in "-O2" "-O2 -mbmi2" "-O3 -march=native"; do
    g++ $fl -std=c++17 -Igaussdca.jl_amd/csrc tests/sanitize/inflate_check.cpp $D/inflate_$v.cpp -o $D/ic -lz || continue
    echo "== $v [$fl]"; taskset -c 3 $D/ic files $D/fam.fasta.gz $D/famC.fasta.gz tests/golden/reference/large.fasta.gz
  done
done
rm -rf $D
