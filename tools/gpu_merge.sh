#!/bin/bash
# First check of the merged sweep: probe (bit-identity + times) and the phased / schedule tests.
#   gpurun --timeout 1500 -- 'bash tools/gpu_merge.sh <tag>'
tag=${1:-merge}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
timeout 300 python tools/merge_probe.py --ks 1 2 4 8 > $out/probe_B.log 2>&1; echo "rc $?" >> $out/probe_B.log; cat $out/probe_B.log
timeout 300 python tools/merge_probe.py --sizes 100:6000 160:9000 240:12000 300:8000 200:20000 130:7000 --theta -1 --ks 2 4 6 8 > $out/probe_mixed.log 2>&1; echo "rc $?" >> $out/probe_mixed.log; cat $out/probe_mixed.log
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "phase_batched or schedule or watchdog or election" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log ); tail -5 $out/pytest.log
