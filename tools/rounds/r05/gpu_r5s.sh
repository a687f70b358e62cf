#!/bin/bash
# round 5, call s: streams of the side-by-side front ends (GDCA_PHASED_STREAMS)
out=gpurun_out/r5s; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "phase or merged or batch" > $out/pytest_phased.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest_phased.log
for rep in 1 2; do for w in 64 4 3 2 1; do
  GDCA_PHASED_STREAMS=$w timeout 300 python bench.py --config B --pipeline 8 --phased --steps 80 --warmup 16 --no-cpu-baseline --no-other-configs > $out/B_m8_w$w.$rep.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/B_m8_w$w.$rep.json').read().strip().splitlines()[-1]); print('B merged8 streams $w rep $rep', round(d['value'],1))"
done; done
for w in 64 4 2; do for P in 8 16; do
  GDCA_PHASED_STREAMS=$w timeout 600 python bench.py --config E --pipeline $P --phased --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/E_phased${P}_w$w.json 2>> $out/err.log
  python -c "import json; d=json.loads(open('$out/E_phased${P}_w$w.json').read().strip().splitlines()[-1]); print('E phased$P streams $w', round(d['value'],2))"
done; done
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_B_merged8 -- python3 $R/bench.py --config B --pipeline 8 --phased --steps 40 --no-cpu-baseline > $R/$out/prof_B_merged8.log 2>&1 < /dev/null
