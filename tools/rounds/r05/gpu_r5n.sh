#!/bin/bash
out=gpurun_out/r5n; mkdir -p $out
for o in 0 1; do for n in 4000 6000; do
  GDCA_CHAIN_ORDER=$o GDCA_SWEEP_TRACE=$out/trace_${n}_o$o.txt timeout 300 python tools/sweep_trace.py $n 99 > /dev/null 2>&1
done; done
ls -la $out
