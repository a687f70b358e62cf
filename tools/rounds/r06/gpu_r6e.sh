#!/bin/bash
# Round 6: the whole E batch under every schedule (one GPU), then the default bench run with E256 in it
out=gpurun_out/r6e; mkdir -p $out
python -c "import torch" 2>/dev/null
SECONDS=0
for sched in "--pipeline 1" "--pipeline 2" "--pipeline 8 --phased" "--pipeline 16 --phased" "--pipeline 32 --phased"; do
  tag=$(echo $sched | tr -d ' -')
  timeout 900 python bench.py --config E --no-cpu-baseline --steps 1 --warmup 1 $sched > $out/E256_$tag.json 2> $out/E256_$tag.err
  python - $out/E256_$tag.json "E256 $sched" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.2f families/s  stage_ms %s' % (sys.argv[2], d['value'], {k: round(v, 3) for k, v in d.get('stage_ms', {}).items()}))
except Exception as e:
    print(sys.argv[2], 'unreadable', e)
PY
  echo "  elapsed so far $SECONDS s"
done
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - $out/bench_default.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('C: %.2f ms/step, k_sweep %.3f ms frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
for k, v in d['other_configs'].items():
    print(k, v.get('value'), v.get('unit'), v.get('error', ''))
PY
