#!/bin/bash
tag=${1:-r5f}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_conditioning.py -k "schedule or watchdog or election or residual or merged or phase or masked or inverse or hamming" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -3 $out/pytest_gpu.log
timeout 400 python tools/stress_merged.py --rounds 25 --seed 12 > $out/stress_merged.log 2>&1; tail -1 $out/stress_merged.log
timeout 300 python tools/stress_inverse.py --repeat 2 --sizes 128 384 2560 7424 9000 10000 11600 > $out/stress_inverse.log 2>&1; tail -1 $out/stress_inverse.log
for i in 1 2; do for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done; done 2>&1 | tee $out/ab.log
unset GDCA_LIB
timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B merged8 main value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})" | tee -a $out/ab.log
timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D main k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})" | tee -a $out/ab.log
for n in 10000; do
  GDCA_SWEEP_TRACE=$out/trace_$n.txt timeout 300 python tools/sweep_trace.py $n 99 > $out/trace_$n.log 2>&1
  grep "^# main\|^# shader" $out/trace_$n.log | cut -c1-400
done
rm -f $out/trace_*.txt
for p in "" "--pipeline 2" "--pipeline 8 --phased"; do
  timeout 600 python bench.py --config E --no-cpu-baseline $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E256 [$p] value %.2f families/s' % d['value'], {k: round(x,3) for k,x in d['stage_ms'].items()})" | tee -a $out/ab.log
done
