#!/bin/bash
# Round 5, third call: the late variant reported step by step; the whole GPU suite on the product build (Hamming bound form as list +
# refine kernel, ppb fix); stage times of B merged8 and C against round 4's library.
tag=${1:-r5c}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
for v in late r04late; do
  GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so timeout 150 python tools/late_probe.py > $out/late_probe_$v.log 2>&1; echo "== late_probe $v rc $?"; grep trial $out/late_probe_$v.log | cut -c1-200
done
( timeout 1700 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=6 --deselect tests/test_gpu_conditioning.py > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -12 $out/pytest_gpu.log
for i in 1 2; do for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()}, [(h['kernel'][:12], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
  timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B merged8 $v value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done; done 2>&1 | tee $out/ab.log
unset GDCA_LIB
timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D main k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})" | tee -a $out/ab.log
