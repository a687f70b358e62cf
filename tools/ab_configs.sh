#!/bin/bash
# A/B of library builds over several bench configurations inside ONE GPU-box call:
#   tools/ab_configs.sh "<variant> ..." "<config> ..."      ("main" = the product library; variants from tools/build_variant.sh)
python -c "import torch" 2>/dev/null
for c in $2; do
case $c in B) fl="--steps 50 --warmup 5";; C) fl="--steps 20 --warmup 3";; D) fl="--steps 5 --warmup 2";; E) fl="--steps 2 --warmup 1";; esac
for rep in 1 2; do for v in $1; do
if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
timeout 900 python bench.py --config $c $fl --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%s %-6s value %.3f  ms/step %.3f  inverse %.3f ms  k_sweep %.3f ms at %.3f GHz  frac %.3f' % ('$c', '$v', d['value'], d['ms_per_step'], d['stage_ms']['ms_inverse'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac']))"
done; done; done
