#!/bin/bash
out=gpurun_out/r5zo; mkdir -p $out
timeout 600 python tools/stress_merged.py --rounds 40 > $out/stress_merged.log 2>&1; echo "stress_merged rc $?"; tail -2 $out/stress_merged.log
