#!/bin/bash
# First half of the round's measurements (tests, bench lines of every configuration, kernel statistics); the counter
# passes, traces and micro-benchmarks are tools/gpu_round2.sh:
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_round.sh r03'
# Output under gpurun_out/<tag>/; the summaries are copied into profiles/ by hand afterwards (tools/prof_summary.py).
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
python __graft_entry__.py > $out/entry.log 2>&1; echo "entry rc $?" >> $out/entry.log
( timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=12 > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log )
tail -3 $out/pytest_gpu.log
# bench lines (default flags = the driver's command), then the other configs
timeout 900 python bench.py > $out/bench_C_frob.json 2> $out/bench_C_frob.err
timeout 900 python bench.py --score DI > $out/bench_C_DI.json 2> $out/bench_C_DI.err
timeout 600 python bench.py --config B --steps 50 --warmup 5 > $out/bench_B.json 2> $out/bench_B.err
timeout 900 python bench.py --config D --steps 5 --warmup 2 > $out/bench_D.json 2> $out/bench_D.err
timeout 900 python bench.py --config E --steps 2 --warmup 1 > $out/bench_E.json 2> $out/bench_E.err
timeout 900 python bench.py --config E --steps 2 --warmup 1 --pipeline 2 > $out/bench_E_p2.json 2> $out/bench_E_p2.err
timeout 900 python bench.py --pipeline 4 --phased --no-cpu-baseline > $out/bench_C_p4_phased.json 2> $out/bench_C_p4_phased.err
for f in $out/bench_*.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'inv %.2f ms' % d['stage_ms']['ms_inverse'],
          'roofline %.1f TF (%.3f) at %.3f GHz' % (r['achieved'], r['frac'], r['measured_shader_ghz']),
          'cpu %s on %s threads' % ((d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores')))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
# kernel-trace + stats of the same command (profiled timings are not compared with un-profiled ones)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_frob -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_frob.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_di -- python3 $GRAFT_REPO_ROOT/bench.py --score DI --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_di.log 2>&1
cd $GRAFT_REPO_ROOT
# keep the merged-back directory small: the per-launch CSVs are what the summaries are made from
find $out -name "*.csv" -size +8M -delete
find $out -name "*agent_info*" -delete
du -sh $out
