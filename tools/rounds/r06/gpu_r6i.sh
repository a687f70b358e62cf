#!/bin/bash
# Round 6: the fp4 Hamming form -- parity tests, then C / D / B / E with HAMMING_MODE = bound | mfma | auto
out=gpurun_out/r6i; mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "hamming or config_C_integers or config_D_prop or golden" > $out/pytest_quick.log 2>&1; echo "rc $?" >> $out/pytest_quick.log ); tail -3 $out/pytest_quick.log
for cfg in C D; do
for mode in bound mfma auto; do
  GDCA_HAMMING_MODE=$mode timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 > $out/${cfg}_$mode.json 2> $out/${cfg}_$mode.err
  python - $out/${cfg}_$mode.json "$cfg $mode" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.2f ms/step  stage_ms %s' % (sys.argv[2], d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))
except Exception as e:
    print(sys.argv[2], 'unreadable', e)
PY
done
done
for mode in bound auto; do
  GDCA_HAMMING_MODE=$mode timeout 300 python bench.py --config B --pipeline 8 --phased --no-cpu-baseline --no-other-configs > $out/B8_$mode.json 2> $out/B8_$mode.err
  GDCA_HAMMING_MODE=$mode timeout 600 python bench.py --config E --pipeline 16 --phased --no-cpu-baseline --steps 1 --warmup 1 > $out/E256_$mode.json 2> $out/E256_$mode.err
  python - $out/B8_$mode.json $out/E256_$mode.json "$mode" <<'PY'
import sys, json
for f in sys.argv[1:3]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%s %s: %.2f families/s  stage_ms %s' % (f.split('/')[-1], sys.argv[3], d['value'], {k: round(v, 3) for k, v in d['stage_ms'].items()}))
    except Exception as e:
        print(f, 'unreadable', e)
PY
done
