#!/bin/bash
# round 5, call x: the driver's sequence on the final tree -- smoke, GPU suite, default bench; then the kernel stats of the driver's command
out=gpurun_out/r5x; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5x/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['end_to_end_gdca_sec'], d['stage_ms'])
for k,v in d['other_configs'].items(): print(k, round(v['value'],2), v.get('roofline',{}).get('frac'))
PY
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_frob -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_frob.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_di -- python3 $R/bench.py --score DI --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_di.log 2>&1 < /dev/null
cd $R; python tools/e2e_profile.py C 5 > $out/e2e_C.log 2>&1; cat $out/e2e_C.log | head -6
