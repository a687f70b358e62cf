#!/bin/bash
# Round 6: batched grids on the pipeline's stream; groups side by side (GDCA_PHASED_GRIDS = 0 | 1 | 2 | 4 | -1) at B eight at a time and on the E prefix
out=gpurun_out/r6d; mkdir -p $out
python -c "import torch" 2>/dev/null
for rep in 1 2; do
for g in 0 1 2 -1; do
  GDCA_PHASED_GRIDS=$g timeout 300 python bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs > $out/B8_g${g}_$rep.json 2> $out/B8_g${g}_$rep.err
  python - $out/B8_g${g}_$rep.json "B8 grids=$g" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.1f families/s  ms/step %.3f  stage_ms %s' % (sys.argv[2], d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d.get('stage_ms', {}).items()}))
except Exception as e:
    print(sys.argv[2], 'unreadable', e)
PY
done
done
for P in 8 16; do
for g in 0 1 2 4 -1; do
  GDCA_PHASED_GRIDS=$g timeout 600 python bench.py --config E --families 64 --pipeline $P --phased --no-cpu-baseline > $out/E64_P${P}_g$g.json 2> $out/E64_P${P}_g$g.err
  python - $out/E64_P${P}_g$g.json "E64 phased$P grids=$g" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.2f families/s  stage_ms %s' % (sys.argv[2], d['value'], {k: round(v, 3) for k, v in d.get('stage_ms', {}).items()}))
except Exception as e:
    print(sys.argv[2], 'unreadable', e)
PY
done
done
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "phase or merged or batch" > $out/pytest_quick.log 2>&1; echo "rc $?" >> $out/pytest_quick.log ); tail -3 $out/pytest_quick.log
