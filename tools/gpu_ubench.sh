#!/bin/bash
# Micro-benchmarks behind the design decisions of DESIGN.md 3.1 (dispatch behaviour beside a register-full kernel,
# CU-mask bit -> XCD/SE/CU mapping, hipStreamWaitValue32) and the schedule sweep of the persistent inverse.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_ubench.sh r02'
tag=${1:-rXX}
out=gpurun_out/${tag}_ubench
mkdir -p $out
for b in ubench_insitu ubench_cumap ubench_fit test_waitvalue test_pivot; do
  [ -x tools/_bin/$b ] && timeout 300 tools/_bin/$b > $out/$b.log 2>&1
done
timeout 1000 python tools/sweep_groups.py > $out/sweep_groups.log 2>&1
tail -40 $out/sweep_groups.log
