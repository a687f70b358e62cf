#!/bin/bash
# round 5, call zh: one chain worker per compute unit for groups of two and three + the group rule derived with it: the driver's sequence
out=gpurun_out/r5zh; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?"
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 600 python tools/option_probe.py 200,288,300,320,345,358,380 "GROUP=-1;GROUP=-1,MCU_SOLO=0" 7 > $out/final_rule.log 2>&1; cat $out/final_rule.log
for P in 1 2; do
  timeout 600 python bench.py --config E --pipeline $P --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_p$P.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/E_p$P.json').read().strip().splitlines()[-1]); print('E p$P', round(d['value'],2))"
done
for P in 8 16; do
  timeout 600 python bench.py --config E --pipeline $P --phased --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_phased$P.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/E_phased$P.json').read().strip().splitlines()[-1]); print('E phased$P', round(d['value'],2))"
done
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5zh/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['end_to_end_gdca_sec'])
for k,v in d['other_configs'].items(): print(k, round(v['value'],2), v.get('roofline',{}).get('frac'))
PY
