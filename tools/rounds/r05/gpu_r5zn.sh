#!/bin/bash
out=gpurun_out/r5zn; mkdir -p $out
timeout 900 python tools/stress_inverse.py --sizes 100 129 2560 5000 5700 5800 6016 6100 6500 6800 7000 7424 7700 8320 10000 11600 > $out/stress_inverse.log 2>&1; echo "stress_inverse rc $?"; tail -4 $out/stress_inverse.log
timeout 600 python tools/stress_merged.py --rounds 60 > $out/stress_merged.log 2>&1; echo "stress_merged rc $?"; tail -3 $out/stress_merged.log
timeout 300 python tools/side_by_side_probe.py > $out/side_by_side.log 2>&1; echo "side_by_side rc $?"; tail -3 $out/side_by_side.log
