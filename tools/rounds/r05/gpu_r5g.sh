#!/bin/bash
tag=${1:-r5g}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
GDCA_SWEEP_TIMEOUT_MS=6000 timeout 300 python tools/side_by_side_probe.py 9100 4 > $out/side_by_side.log 2>&1; echo "side_by_side rc $?"; grep -v amdgpu.ids $out/side_by_side.log | cut -c1-300
GDCA_SWEEP_TIMEOUT_MS=6000 timeout 200 python tools/side_by_side_probe.py 2560 8 > $out/side_by_side_B.log 2>&1; echo "side_by_side B rc $?"; grep -v amdgpu.ids $out/side_by_side_B.log | cut -c1-300
one() { # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.1f step %.3f k_sweep %.3f ms %.3f GHz frac %.3f inv %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['stage_ms']['ms_inverse']))"
}
for m in 4 5 6 8 10; do one "C MCUS=$m" GDCA_MCUS=$m -- --steps 20 --warmup 3; done 2>&1 | tee $out/sweeps.log
for g in 3 4; do one "C GROUP=$g" GDCA_GROUP=$g -- --steps 20 --warmup 3; done 2>&1 | tee -a $out/sweeps.log
for t in 256 768 1500; do one "C REM_TAIL=$t" GDCA_REM_TAIL=$t -- --steps 20 --warmup 3; done 2>&1 | tee -a $out/sweeps.log
for m in 2 3 4 6; do one "B merged8 MERGE_MCUS=$m" GDCA_MERGE_MCUS=$m -- --config B --pipeline 8 --phased --steps 80; done 2>&1 | tee -a $out/sweeps.log
one "B merged8 fronts serial" GDCA_PHASED_FRONTS=0 -- --config B --pipeline 8 --phased --steps 80 2>&1 | tee -a $out/sweeps.log
one "B merged8 MERGE_GROUP=1" GDCA_MERGE_GROUP=1 -- --config B --pipeline 8 --phased --steps 80 2>&1 | tee -a $out/sweeps.log
one "B pipeline 4 (no phases)" A=1 -- --config B --pipeline 4 --steps 80 2>&1 | tee -a $out/sweeps.log
for m in 2 4 8; do one "D MCUS=$m" GDCA_MCUS=$m -- --config D --steps 4 --warmup 1; done 2>&1 | tee -a $out/sweeps.log
