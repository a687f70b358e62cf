#!/bin/bash
# The multi-threaded single-member decoder (gdca_gunzip_parallel) on the host of the GPU box: phases and rate by thread count,
# then the reader's phase trace on the same file.   bash tools/experiments/inflate/parallel.sh
cd $(dirname $0)/../../..
D=/tmp/gdca_inflate_par; rm -rf $D; mkdir -p $D
gaussdca.jl_amd/gdca_cli --synth 500 50000 50432 $D/famC.fasta.gz
gaussdca.jl_amd/gdca_cli --synth 1000 100000 856064 $D/famD.fasta.gz
g++ -O2 -std=c++17 -pthread -Igaussdca.jl_amd/csrc tests/sanitize/inflate_check.cpp gaussdca.jl_amd/csrc/gdca_inflate.cpp -o $D/ic -lz || exit 1
GDCA_INFLATE_TRACE=1 $D/ic files $D/famC.fasta.gz $D/famD.fasta.gz 2>&1
python tools/parse_trace.py 2>&1 | grep -A4 "fasta.gz, GDCA_FASTA_THREADS=(default)\|fasta.gz, GDCA_FASTA_THREADS=16" | cut -c1-260
rm -rf $D
