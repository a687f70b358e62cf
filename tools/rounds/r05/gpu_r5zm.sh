#!/bin/bash
out=gpurun_out/r5zm; mkdir -p $out
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5zm/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['end_to_end_gdca_sec'])
for k,v in d['other_configs'].items(): print(k, round(v['value'],2), v.get('roofline',{}).get('frac'))
PY
timeout 600 python bench.py --config E --pipeline 16 --phased --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_phased16.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/E_phased16.json').read().strip().splitlines()[-1]); print('E phased16', round(d['value'],2))"
timeout 600 python bench.py --config E --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_p1.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/E_p1.json').read().strip().splitlines()[-1]); print('E p1', round(d['value'],2))"
