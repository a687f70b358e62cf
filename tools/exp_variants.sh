#!/bin/bash
# Compare builds of libgdca.so on the GPU box (kernel experiments): tools/exp_variants.sh <tag> <variant> [<variant> ...]
# variant "main" = gaussdca.jl_amd/libgdca.so, anything else = gaussdca.jl_amd/libgdca_<variant>.so (tools/build_variant.sh).
# Per variant: residual / determinism check on a few sizes, the in-kernel trace of one inverse at n = 10 000 and 20 000
# (tile-item microseconds, shader clock), and the bench line of config C (HIP-event time of k_sweep).
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
for spec in "$@"; do
  # <variant>[@ENV=VAL[,ENV=VAL...]]: environment switches of the library for this run
  v=${spec%%@*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*@}
  if [ "$v" = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  for kv in ${envs//,/ }; do export "$kv"; done
  v=${spec//[@=,]/_}
  echo "=== $v"
  timeout 300 python tools/stress_inverse.py --repeat 2 --sizes 384 2560 7424 9000 10000 11600 > $out/stress_$v.log 2>&1; tail -1 $out/stress_$v.log
  for n in 10000 20000; do
    GDCA_SWEEP_TRACE=$out/trace_${v}_$n.txt timeout 300 python tools/sweep_trace.py $n 99 > $out/trace_${v}_$n.log 2>&1
    grep "^# main\|^# shader" $out/trace_${v}_$n.log | cut -c1-330
  done
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_C_$v.json 2> $out/bench_C_$v.err
  python - $out/bench_C_$v.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('bench C: value %.3f ms/step %.2f k_sweep %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
except Exception as e:
    print('bench unreadable', e)
PY
  for kv in ${envs//,/ }; do unset "${kv%%=*}"; done
done
rm -f $out/trace_*.txt
