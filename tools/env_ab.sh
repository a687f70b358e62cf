#!/bin/bash
# A/B of environment switches of the library inside ONE GPU-box call:
#   tools/env_ab.sh "<config> ..." "VAR=val[,VAR=val]" "VAR=val" ...      ("-" = no switch)
python -c "import torch" 2>/dev/null
cfgs=$1; shift
for c in $cfgs; do
case $c in B) fl="--steps 50 --warmup 5";; C) fl="--steps 20 --warmup 3";; D) fl="--steps 5 --warmup 2";; E) fl="--steps 2 --warmup 1";; esac
for rep in 1 2; do for v in "$@"; do
( if [ "$v" != "-" ]; then for kv in ${v//,/ }; do export $kv; done; fi
timeout 900 python bench.py --config $c $fl --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%s %-34s value %.3f  ms/step %.3f  inverse %.3f ms  k_sweep %.3f ms at %.3f GHz  frac %.3f' % ('$c', '$v', d['value'], d['ms_per_step'], d['stage_ms']['ms_inverse'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac']))" )
done; done; done
