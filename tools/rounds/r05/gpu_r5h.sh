#!/bin/bash
tag=${1:-r5h}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
( timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_conditioning.py -k "fn or FN or golden or fused or parity or config_C or config_B or merged or phase or stepwise or side_by_side" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -4 $out/pytest_gpu.log
GDCA_SWEEP_TIMEOUT_MS=8000 timeout 300 python tools/side_by_side_probe.py 9100 4 merged > $out/side_by_side_merged.log 2>&1; echo "merged beside big rc $?"; grep -v amdgpu.ids $out/side_by_side_merged.log | cut -c1-300
for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  GDCA_SWEEP_TIMEOUT_MS=8000 timeout 300 python tools/side_by_side_probe.py 9100 4 merged 2>&1 | grep -v amdgpu.ids | cut -c1-300 | sed "s/^/$v: /"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C $v k_sweep %.3f ms frac %.3f step %.2f' % (r['avg_launch_ms'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()}, [(h['kernel'][:6], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
  timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B merged8 $v value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done 2>&1 | tee $out/ab.log
unset GDCA_LIB
timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D main k_sweep %.3f ms frac %.3f step %.2f' % (r['avg_launch_ms'], r['frac'], d['ms_per_step']), [(h['kernel'][:6], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])" | tee -a $out/ab.log
timeout 400 python tools/stress_merged.py --rounds 15 --seed 21 > $out/stress_merged.log 2>&1; tail -1 $out/stress_merged.log
