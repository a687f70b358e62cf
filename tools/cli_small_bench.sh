#!/bin/bash
# gdca_cli --batch on F small families (config B's size: N = 128, M = 10 000), merged batches against one launch per family:
#   bash tools/cli_small_bench.sh [F] [scratch dir]
F=${1:-96}
D=${2:-/tmp/gdca_cli_small}
rm -rf $D; mkdir -p $D/in $D/out
for f in $(seq 1 $F); do echo 128 10000 $((0xB128 + f)) $D/in/fam$(printf %03d $f).fasta; done | xargs -P 16 -L 1 gaussdca.jl_amd/gdca_cli --synth > /dev/null
for m in 1 8 1 8; do
  echo "--merge $m: $(gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --gpus 1 --inflight 2 --merge $m 2>&1 | tail -3 | head -2 | tr '\n' ' ')"
done
rm -rf $D
