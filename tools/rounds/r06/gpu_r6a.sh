#!/bin/bash
# Round 6, first box call: the batched grids (GDCA_PHASED_GRIDS) against the per-member launches, then the whole GPU suite.
#   gpurun --timeout 2400 -- 'bash tools/rounds/r06/gpu_r6a.sh'
out=gpurun_out/r6a; mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "phase or merged or batch or smoke or golden" > $out/pytest_quick.log 2>&1; echo "rc $?" >> $out/pytest_quick.log ); tail -3 $out/pytest_quick.log
for rep in 1 2 3; do
  for g in 0 1; do
    GDCA_PHASED_GRIDS=$g timeout 300 python bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs > $out/B8_g${g}_$rep.json 2> $out/B8_g${g}_$rep.err
    python - $out/B8_g${g}_$rep.json $g <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('B merged8 grids=%s: %.1f families/s  ms/step %.3f  stage_ms %s' % (sys.argv[2], d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d.get('stage_ms', {}).items()}))
except Exception as e:
    print('unreadable', e)
PY
  done
done
for g in 0 1; do
  GDCA_PHASED_GRIDS=$g timeout 600 python bench.py --config E --families 64 --pipeline 8 --phased --no-cpu-baseline > $out/E64_g$g.json 2> $out/E64_g$g.err
  python - $out/E64_g$g.json $g <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('E64 phased8 grids=%s: %.2f families/s' % (sys.argv[2], d['value']))
except Exception as e:
    print('unreadable', e)
PY
done
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest_all.log 2>&1; echo "rc $?" >> $out/pytest_all.log ); tail -3 $out/pytest_all.log
