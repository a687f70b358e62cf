#!/bin/bash
# round 5, call zi: one chain worker per compute unit in MERGED launches (experiment)
out=gpurun_out/r5zi; mkdir -p $out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config B --pipeline 8 --phased --steps 80 --warmup 16 --no-cpu-baseline --no-other-configs > $out/x.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/x.json').read().strip().splitlines()[-1]); print('B merged8 $label', round(d['value'],1), 'inv', round(d['stage_ms']['ms_inverse_update'],4))"
}
for rep in 1 2; do
  run default X=1
  run solo4 GDCA_MCU_SOLO=1 GDCA_MERGE_MCUS=4
  run solo6 GDCA_MCU_SOLO=1 GDCA_MERGE_MCUS=6
  run solo8 GDCA_MCU_SOLO=1 GDCA_MERGE_MCUS=8
  run solo3 GDCA_MCU_SOLO=1 GDCA_MERGE_MCUS=3
done
timeout 300 python tools/stress_merged.py --batches 60 > $out/stress.log 2>&1; tail -2 $out/stress.log
GDCA_MCU_SOLO=1 GDCA_MERGE_MCUS=6 timeout 300 python tools/stress_merged.py --batches 60 > $out/stress_solo.log 2>&1; tail -2 $out/stress_solo.log
