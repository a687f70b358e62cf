#!/bin/bash
# XCD-aware hand-out of the remainder tiles: correctness, then A/B by option inside one call
tag=${1:-r5k}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
timeout 300 python tools/stress_inverse.py --repeat 2 --sizes 128 384 1100 2560 4000 6000 7424 9000 10000 11600 > $out/stress_inverse.log 2>&1; tail -1 $out/stress_inverse.log
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_conditioning.py -k "schedule or watchdog or election or residual or merged or phase or masked or inverse" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -3 $out/pytest_gpu.log
timeout 300 python tools/stress_merged.py --rounds 12 --seed 31 > $out/stress_merged.log 2>&1; tail -1 $out/stress_merged.log
one() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.2f step %.3f k_sweep %.3f ms %.3f GHz frac %.3f inv %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['stage_ms']['ms_inverse']))"; }
for i in 1 2; do
  one "C XCD=0" GDCA_XCD=0 -- --steps 20 --warmup 3
  one "C XCD=1 sb4" GDCA_XCD=1 -- --steps 20 --warmup 3
  one "C XCD=1 sb8" GDCA_XCD=1 GDCA_XCD_SB=8 -- --steps 20 --warmup 3
  one "C XCD=1 sb2" GDCA_XCD=1 GDCA_XCD_SB=2 -- --steps 20 --warmup 3
done 2>&1 | tee $out/xcd_ab.log
for x in "GDCA_XCD=0" "GDCA_XCD=1" "GDCA_XCD=1 GDCA_XCD_SB=4" "GDCA_XCD=1 GDCA_XCD_SB=16"; do
  one "N300 $x" $x -- --N 300 --M 8000 --steps 30 --warmup 3
  one "N200 $x" $x -- --N 200 --M 8000 --steps 40 --warmup 3
  one "N400 $x" $x -- --N 400 --M 8000 --steps 20 --warmup 3
  one "B $x" $x -- --config B --steps 40 --warmup 5
done 2>&1 | tee -a $out/xcd_ab.log
for x in "GDCA_XCD=0" "GDCA_XCD=1" "GDCA_XCD=1 GDCA_XCD_SB=8"; do
  one "D $x" $x -- --config D --steps 4 --warmup 1
  one "B merged8 $x" $x -- --config B --pipeline 8 --phased --steps 80
  one "E64 phased8 $x" $x -- --config E --families 64 --pipeline 8 --phased
done 2>&1 | tee -a $out/xcd_ab.log
