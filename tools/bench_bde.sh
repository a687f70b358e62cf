python -c "import torch" 2>/dev/null
mkdir -p gpurun_out/r03a
for c in "B --steps 50 --warmup 5" "D --steps 5 --warmup 2" "E --steps 2 --warmup 1"; do
set -- $c
timeout 900 python bench.py --config $c --no-cpu-baseline > gpurun_out/r03a/bench_$1.json 2> gpurun_out/r03a/bench_$1.err
python - gpurun_out/r03a/bench_$1.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3f ms/step %.2f' % (d['value'], d['ms_per_step']), 'stage', {k: round(v, 2) for k, v in d['stage_ms'].items()}, 'k_sweep %.3f ms %.3f GHz frac %.3f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac']))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
