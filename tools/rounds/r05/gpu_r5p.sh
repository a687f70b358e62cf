#!/bin/bash
# round 5, call p: the exact Meff kernel -- full GPU suite, smoke, default bench, kernel stats
out=gpurun_out/r5p; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?"
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5p/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['stage_ms'])
for k,v in d['other_configs'].items(): print(k, v['value'], v.get('roofline',{}).get('frac'), v['stage_ms']['ms_weights'])
PY
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_frob -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_frob.log 2>&1 < /dev/null
cd $R; f=$(find $out/prof_frob -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150; grep k_meff $f | cut -c1-150
