#!/bin/bash
# Round 5, first call: the whole GPU suite (conditioning window included, its log kept), bench C with stage times.
tag=${1:-r5a}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 1500 python -m pytest tests/test_gpu_conditioning.py -m gpu -q -x -s -p no:cacheprovider > $out/conditioning.log 2>&1; echo "rc $?" >> $out/conditioning.log ) < /dev/null
grep "forward error\|kappa_1\|passed\|failed\|rc " $out/conditioning.log | cut -c1-330
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 --deselect tests/test_gpu_conditioning.py > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -12 $out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 > $out/bench_C.json 2> $out/bench_C.err
python - $out/bench_C.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('bench C: value %.3f ms/step %.2f k_sweep %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']), d['stage_ms'])
except Exception as e:
    print('bench unreadable', e)
PY
