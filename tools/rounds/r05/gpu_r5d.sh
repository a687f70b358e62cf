#!/bin/bash
tag=${1:-r5d}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
for v in late main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 60 python tools/late_probe2.py > $out/late_probe2_$v.log 2>&1; echo "== late_probe2 $v rc $?"; grep -v amdgpu.ids $out/late_probe2_$v.log | cut -c1-200
done
unset GDCA_LIB
( timeout 1500 python -m pytest tests/test_gpu_conditioning.py -m gpu -q -x -s -p no:cacheprovider > $out/conditioning.log 2>&1; echo "rc $?" >> $out/conditioning.log ) < /dev/null
grep "passed\|failed\|rc \|pc=0.8\|pc=0.5" $out/conditioning.log | cut -c1-300
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_conditioning.py -k "hamming or neighbour or weights or phase or merged or golden or config_B or config_C or fused or stepwise or schedule" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -5 $out/pytest_gpu.log
for i in 1 2; do for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()}, [(h['kernel'][:12], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
  timeout 300 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B merged8 $v value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
  timeout 300 python bench.py --config B --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $v value %.1f step %.3f' % (d['value'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done; done 2>&1 | tee $out/ab.log
for v in r04 main; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --config E --families 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E64 $v value %.2f' % d['value'], {k: round(x,3) for k,x in d['stage_ms'].items()})"
  timeout 300 python bench.py --config E --families 64 --no-cpu-baseline --pipeline 8 --phased 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E64 phased8 $v value %.2f' % d['value'], {k: round(x,3) for k,x in d['stage_ms'].items()})"
  timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('D $v k_sweep %.3f ms %.3f GHz frac %.3f step %.2f' % (r['avg_launch_ms'], r['measured_shader_ghz'], r['frac'], d['ms_per_step']), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done 2>&1 | tee -a $out/ab.log
