#!/bin/bash
# The whole GPU suite, then the bench lines of C (headline) and B; everything under gpurun_out/<tag>/.
#   gpurun --timeout 2400 -- 'bash tools/gpu_full.sh <tag>'
tag=${1:-full}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log ); tail -6 $out/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_C.json 2> $out/bench_C.err
timeout 300 python bench.py --no-cpu-baseline --config B > $out/bench_B.json 2> $out/bench_B.err
timeout 300 python bench.py --no-cpu-baseline --config B --pipeline 4 --phased > $out/bench_B_p4.json 2> $out/bench_B_p4.err
python - $out <<'PY'
import sys, json
for name in ("bench_C", "bench_B", "bench_B_p4"):
    try:
        d = json.loads(open("%s/%s.json" % (sys.argv[1], name)).read().strip().splitlines()[-1])
        print('%s: value %.3f ms/step %.3f k_sweep %.3f ms frac %.3f stage %s' % (name, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))
    except Exception as e:
        print(name, 'unreadable', e)
PY
