#!/bin/bash
# A/B of the single-block chain options inside ONE GPU-box call (box-to-box differences are as large as some effects):
# GDCA_SLAB (row-slab items on the chain | panel and tile items) x GDCA_RING (Pg / panel buffers) on configs B and E.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/slab_ab.sh'
python -c "import torch" 2>/dev/null
for rep in 1 2; do for v in "0 2" "1 2" "0 8" "1 8"; do set -- $v
for c in "B --steps 50 --warmup 5" "E --steps 2 --warmup 1"; do
GDCA_SLAB=$1 GDCA_RING=$2 timeout 900 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('slab $1 ring $2 config ${c%% *}: value %.3f  ms/step %.3f  inverse %.3f ms  k_sweep %.3f ms at %.3f GHz  frac %.3f' % (d['value'], d['ms_per_step'], d['stage_ms']['ms_inverse'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac']))"
done; done; done
