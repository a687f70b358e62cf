#!/bin/bash
out=gpurun_out/r5u; mkdir -p $out
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $R/$out/prof -- python3 $R/bench.py --config B --pipeline 8 --phased --steps 64 --warmup 16 --no-cpu-baseline > $R/$out/prof.log 2>&1 < /dev/null
cd $R; ls -la $out/prof/*/ | head; 
