#!/bin/bash
tag=${1:-r5e}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_conditioning.py -k "hamming or neighbour or weights or phase or merged or config_B or batch_driver" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -4 $out/pytest_gpu.log
for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
  timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B merged8 $v value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
  timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done 2>&1 | tee $out/ab.log
unset GDCA_LIB
for n in 10000 20000; do
  GDCA_SWEEP_TRACE=$out/trace_$n.txt timeout 300 python tools/sweep_trace.py $n 99 > $out/trace_$n.log 2>&1
  grep "^# main\|^# shader\|^# pivot" $out/trace_$n.log | cut -c1-400
done
rm -f $out/trace_*.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_frob -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_frob.log 2>&1 < /dev/null
cd $R
find $out -name "*kernel_stats.csv" | head -2 | while read f; do head -25 "$f" | cut -d, -f1-8; done
find $out -name "*.csv" -size +8M -delete; find $out -name "*agent_info*" -delete
