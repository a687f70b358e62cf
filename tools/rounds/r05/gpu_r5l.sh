#!/bin/bash
# round 5, call l: final validation of the tree + traces of the mid sizes
out=gpurun_out/r5l; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?"
for n in 4000 6000 8000; do
  GDCA_SWEEP_TRACE=$out/trace_$n.txt timeout 300 python tools/sweep_trace.py $n 5,6,20 > $out/trace_$n.log 2>&1
  grep "^# nblk\|^# main\|^# shader\|^# pivot\|^M list" $out/trace_$n.log | cut -c1-400
done
rm -f $out/trace_*.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic_source'))
for k,v in d['other_configs'].items(): print(k, v['value'], v.get('roofline',{}).get('frac'))
PY
