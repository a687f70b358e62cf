#!/bin/bash
out=gpurun_out/r5r; mkdir -p $out
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_B_merged8 -- python3 $R/bench.py --config B --pipeline 8 --phased --steps 40 --no-cpu-baseline > $R/$out/prof_B_merged8.log 2>&1 < /dev/null
cd $R; find $out -name "*kernel_stats.csv" | head -2
