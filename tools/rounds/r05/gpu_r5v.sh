#!/bin/bash
out=gpurun_out/r5v; mkdir -p $out
INFLIGHT="1 2 3" timeout 900 bash tools/cli_batch_bench.sh 128 > $out/cli_batch_128.log 2>&1; grep "batch:\|steady" $out/cli_batch_128.log
timeout 600 python bench.py --config E --families 128 --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E128_p1.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/E128_p1.json').read().strip().splitlines()[-1]); print('bench E128 p1', round(d['value'],2))"
timeout 600 python bench.py --config E --families 128 --pipeline 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E128_p2.json 2>> $out/err.log
python -c "import json; d=json.loads(open('$out/E128_p2.json').read().strip().splitlines()[-1]); print('bench E128 p2', round(d['value'],2))"
