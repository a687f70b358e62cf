#!/bin/bash
tag=${1:-r5j}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch" 2>/dev/null
for i in 1 2; do for v in main fn12 fn12w4 fn6w8; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C $v step %.2f score %.3f' % (d['ms_per_step'], d['stage_ms']['ms_score']), [(h['kernel'][:6], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
done; done 2>&1 | tee $out/fn_ab.log
for v in main fn12 fn12w4 fn6w8; do
  if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
  timeout 300 python bench.py --config D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('D $v', [(h['kernel'][:6], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
  timeout 300 python bench.py --config B --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $v', [(h['kernel'][:6], round(h['avg_launch_ms'],4), round(h['frac'],3)) for h in d.get('roofline_hbm', [])])"
done 2>&1 | tee -a $out/fn_ab.log
unset GDCA_LIB
( timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "side_by_side" > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -3 $out/pytest_gpu.log
