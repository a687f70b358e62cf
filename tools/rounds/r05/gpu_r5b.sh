#!/bin/bash
# Round 5, second call: the restructured sweep kernels (cold items out of line, no spills in the chunk loops).
#  1. the round-4 "written late" variant of the merged loop on round 4's code (expected: wrong inverses) and on the new structure
#  2. product build: schedule / watchdog / election / merged tests, stress ladders
#  3. A/B against round 4's library inside this one box call
tag=${1:-r5b}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
for v in r04late late main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 400 python tools/stress_merged.py --rounds 25 --seed 11 > $out/stress_merged_$v.log 2>&1; echo "stress_merged $v: rc $? $(tail -1 $out/stress_merged_$v.log | cut -c1-200)"
done
unset GDCA_LIB
timeout 300 python tools/stress_inverse.py --repeat 2 --sizes 128 384 2560 7424 9000 10000 11600 > $out/stress_inverse.log 2>&1; tail -1 $out/stress_inverse.log
( timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "schedule or watchdog or election or residual or merged or phase_batched or masked or inverse" --deselect tests/test_gpu_conditioning.py > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log ); tail -4 $out/pytest.log
bash tools/ab_bench.sh 3 r04 main 2>&1 | tee $out/ab_bench.log
for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), d['stage_ms'])"
  timeout 300 python bench.py --config B --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('B $v k_sweep %.3f ms frac %.3f step %.3f' % (r['avg_launch_ms'], r['frac'], d['ms_per_step']))"
  timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('B merged8 $v value %.1f step %.3f' % (d['value'], d['ms_per_step']))"
done 2>&1 | tee $out/ab_others.log
