#!/bin/bash
out=gpurun_out/r6f; mkdir -p $out
python -c "import torch" 2>/dev/null
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 > $out/C_$i.json 2> $out/C_$i.err
python - $out/C_$i.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('C: %.2f ms/step  stage_ms %s k_sweep frac %.3f' % (d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms'].items()}, d['roofline']['frac']))
PY
done
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "hamming or phase or config_C" > $out/pytest_quick.log 2>&1; echo "rc $?" >> $out/pytest_quick.log ); tail -3 $out/pytest_quick.log
