#!/bin/bash
# A/B of library builds inside ONE GPU-box call (box-to-box and run-to-run clock differences are as large as the effects
# being measured): tools/ab_bench.sh <rounds> <variant> [<variant> ...]; "main" = the product library.
# Prints k_sweep ms, the clock the kernel measured, and ms normalised to 2.1 GHz.
rounds=${1:-3}; shift
python -c "import torch" 2>/dev/null
for i in $(seq $rounds); do for v in "$@"; do
if [ $v = main ]; then unset GDCA_LIB; else export GDCA_LIB=$PWD/gaussdca.jl_amd/libgdca_$v.so; fi
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-6s k_sweep %.3f ms  %.3f GHz  -> %.3f ms @2.1GHz   step %.2f ms  frac %.3f (%.3f of attainable)' % ('$v', r['avg_launch_ms'], r['measured_shader_ghz'], r['avg_launch_ms']*r['measured_shader_ghz']/2.1, d['ms_per_step'], r['frac'], r['frac_of_attainable_at_measured_clock']))"
done; done
