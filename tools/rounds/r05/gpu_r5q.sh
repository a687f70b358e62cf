#!/bin/bash
# round 5, call q: exact Meff, no side stream -- full GPU suite and the batch lines
out=gpurun_out/r5q; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
for rep in 1 2 3; do
  timeout 300 python bench.py --config B --pipeline 8 --phased --steps 80 --warmup 16 --no-cpu-baseline --no-other-configs > $out/B_m8.$rep.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/B_m8.$rep.json').read().strip().splitlines()[-1]); print('B merged8 rep $rep', round(d['value'],1), d['stage_ms'])"
done
timeout 300 python bench.py --config B --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $out/B.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/B.json').read().strip().splitlines()[-1]); print('B', round(d['ms_per_step'],4), d['stage_ms'])"
for P in 1 2; do
  timeout 600 python bench.py --config E --pipeline $P --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_p$P.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/E_p$P.json').read().strip().splitlines()[-1]); print('E p$P', round(d['value'],2))"
done
for P in 8 16; do
  timeout 600 python bench.py --config E --pipeline $P --phased --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_phased$P.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/E_phased$P.json').read().strip().splitlines()[-1]); print('E phased$P', round(d['value'],2))"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $out/C.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/C.json').read().strip().splitlines()[-1]); print('C', round(d['ms_per_step'],3), d['stage_ms'])"
timeout 300 python bench.py --config D --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $out/D.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/D.json').read().strip().splitlines()[-1]); print('D', round(d['ms_per_step'],3), d['roofline']['frac'], d['stage_ms'])"
[ -f tools/cli_batch_bench.sh ] && timeout 600 bash tools/cli_batch_bench.sh > $out/cli_batch.log 2>&1; tail -5 $out/cli_batch.log
