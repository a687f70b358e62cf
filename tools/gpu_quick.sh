#!/bin/bash
# Quick GPU check after a kernel change: the inverse's schedule / watchdog tests, the stress ladder, bench C.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_quick.sh <tag> [pytest -k expression]'
tag=${1:-q}; sel=${2:-"schedule or watchdog or election or residual"}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$sel" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log ); tail -4 $out/pytest.log
timeout 300 python tools/stress_inverse.py --repeat 2 --sizes 128 384 2560 7424 9000 10000 11600 > $out/stress.log 2>&1; tail -1 $out/stress.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_C.json 2> $out/bench_C.err
python - $out/bench_C.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('bench C: value %.3f ms/step %.2f k_sweep %.3f ms frac %.3f issued/alg %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_flops_issued_per_launch'] / d['roofline']['flops_per_launch']))
except Exception as e:
    print('bench unreadable', e)
PY
