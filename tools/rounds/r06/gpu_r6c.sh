#!/bin/bash
# Round 6: the same eight (sixteen) families of the E batch one after the other and as one phase batch with batched grids: per-kernel times
out=gpurun_out/r6c; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
for nf in 8 16; do
  rocprofv3 --kernel-trace -d $out/seq$nf -o s -- python3 bench.py --config E --families $nf --no-cpu-baseline --steps 1 --warmup 1 > $out/seq$nf.json 2> $out/seq$nf.err
  GDCA_PHASED_GRIDS=1 rocprofv3 --kernel-trace -d $out/g1_$nf -o g -- python3 bench.py --config E --families $nf --pipeline $nf --phased --phased-one-set --no-cpu-baseline --steps 1 --warmup 1 > $out/g1_$nf.json 2> $out/g1_$nf.err
done
ls -la $out/*
