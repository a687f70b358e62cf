#!/bin/bash
# Round 6: the persistent three-plane Hamming form of short alignments (GDCA_HAMMING_SHORT = 0 | 1)
out=gpurun_out/r6l; mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "hamming or phase or config_B or golden or config_E_fam" > $out/pytest_quick.log 2>&1; echo "rc $?" >> $out/pytest_quick.log ); tail -3 $out/pytest_quick.log
for rep in 1 2; do
for sh in 0 1; do
  GDCA_HAMMING_SHORT=$sh timeout 300 python bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs > $out/B8_s${sh}_$rep.json 2> $out/B8_s${sh}_$rep.err
  GDCA_HAMMING_SHORT=$sh timeout 300 python bench.py --config B --no-cpu-baseline --no-other-configs > $out/B_s${sh}_$rep.json 2> $out/B_s${sh}_$rep.err
  GDCA_HAMMING_SHORT=$sh timeout 600 python bench.py --config E --pipeline 16 --phased --no-cpu-baseline --steps 1 --warmup 1 > $out/E256_s${sh}_$rep.json 2> $out/E256_s${sh}_$rep.err
  python - $out/B8_s${sh}_$rep.json $out/B_s${sh}_$rep.json $out/E256_s${sh}_$rep.json "short=$sh" <<'PY'
import sys, json
for f in sys.argv[1:4]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%s %s: %.2f families/s  stage_ms %s' % (f.split('/')[-1], sys.argv[4], d['value'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))
    except Exception as e:
        print(f, 'unreadable', e)
PY
done
done
