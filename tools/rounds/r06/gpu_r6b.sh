#!/bin/bash
# Round 6: kernel statistics of a phase batch with and without batched grids (E prefix and B eight at a time)
out=gpurun_out/r6b; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
for g in 0 1; do
  export GDCA_PHASED_GRIDS=$g
  rocprofv3 --kernel-trace --stats -d $out/E32_g$g -o e -- python3 bench.py --config E --families 32 --pipeline 8 --phased --no-cpu-baseline --steps 1 --warmup 1 > $out/E32_g$g.json 2> $out/E32_g$g.err
  rocprofv3 --kernel-trace --stats -d $out/B8_g$g -o b -- python3 bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs --steps 20 --warmup 2 > $out/B8_g$g.json 2> $out/B8_g$g.err
done
find $out -name "*kernel_stats.csv" | head
for f in $(find $out -name "*kernel_stats.csv"); do echo "== $f"; head -25 $f | cut -c1-200; done
# keep only the stats files (the traces are large)
find $out -name "*kernel_trace.csv" -size +20M -delete
