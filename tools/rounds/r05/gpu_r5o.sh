#!/bin/bash
# round 5, call o: hardware queues for the side-by-side front ends (GPU_MAX_HW_QUEUES; the runtime's default is 4)
out=gpurun_out/r5o; mkdir -p $out
for rep in 1 2; do for hq in 4 8 16; do
  GPU_MAX_HW_QUEUES=$hq timeout 300 python bench.py --config B --pipeline 8 --phased --steps 80 --warmup 16 --no-cpu-baseline --no-other-configs > $out/B_m8_hq$hq.$rep.json 2> $out/err.log
  python - <<PY
import json
d=json.loads(open('$out/B_m8_hq$hq.$rep.json').read().strip().splitlines()[-1]); print('B merged8 hq $hq rep $rep', round(d['value'],1), 'fam/s')
PY
done; done
for hq in 4 8 16; do
  GPU_MAX_HW_QUEUES=$hq timeout 600 python bench.py --config E --pipeline 16 --phased --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_p16_hq$hq.json 2>> $out/err.log
  python - <<PY
import json
d=json.loads(open('$out/E_p16_hq$hq.json').read().strip().splitlines()[-1]); print('E phased16 hq $hq', round(d['value'],2), 'fam/s')
PY
done
for hq in 4 8; do
  GPU_MAX_HW_QUEUES=$hq timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $out/C_hq$hq.json 2>> $out/err.log
  python - <<PY
import json
d=json.loads(open('$out/C_hq$hq.json').read().strip().splitlines()[-1]); print('C hq $hq', round(d['ms_per_step'],3), 'ms')
PY
done
