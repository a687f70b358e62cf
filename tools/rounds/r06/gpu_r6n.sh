#!/bin/bash
out=gpurun_out/r6n; mkdir -p $out
python -c "import torch" 2>/dev/null
for mg in -1 1 2 3 4; do
  for mc in -1 2 4; do
  GDCA_MERGE_GROUP=$mg GDCA_MERGE_MCUS=$mc timeout 300 python bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs > $out/B8_g${mg}_m$mc.json 2> $out/B8_g${mg}_m$mc.err
  python - $out/B8_g${mg}_m$mc.json "MERGE_GROUP=$mg MERGE_MCUS=$mc" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.1f families/s  inverse %.3f ms' % (sys.argv[2], d['value'], d['stage_ms']['ms_inverse']))
except Exception as e:
    print(sys.argv[2], 'unreadable', e)
PY
  done
done
