python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "phase_batched or async_pipeline" 2>&1 | tail -3
for a in "--pipeline 1" "--pipeline 2 --phased" "--pipeline 4 --phased" "--pipeline 8 --phased"; do
timeout 300 python bench.py --no-cpu-baseline --steps 24 --warmup 8 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C %-22s value %.2f fam/s  step %.2f ms  k_sweep %.3f ms %.3f GHz frac %.3f' % ('$a', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['measured_shader_ghz'], r['frac']))"
done
for a in "--pipeline 1" "--pipeline 2" "--pipeline 4 --phased" "--pipeline 8 --phased" "--pipeline 16 --phased"; do
timeout 600 python bench.py --config E --no-cpu-baseline --steps 2 --warmup 1 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('E %-22s value %.2f fam/s  k_sweep avg %.3f ms frac %.3f' % ('$a', d['value'], r['avg_launch_ms'], r['frac']))"
done
