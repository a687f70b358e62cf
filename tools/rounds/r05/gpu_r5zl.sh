#!/bin/bash
out=gpurun_out/r5zl; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
