#!/bin/bash
# Copies / summarises what tools/gpu_round6.sh left under gpurun_out/<tag>/ into profiles/<prefix>_* (run here, after the box call):
#   bash tools/collect_round_profiles.sh r06c r06
tag=${1:?tag}; pre=${2:?prefix}
O=gpurun_out/$tag
for f in bench_default bench_B bench_B_merged8 bench_B_merged8_grids0 bench_C_DI bench_D bench_E bench_E_p2 bench_E_phased8 bench_E_phased16 bench_E_phased16_grids0; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json > profiles/${pre}_$f.json
done
cp $O/stress_merged.log profiles/${pre}_stress_merged.log
cp $O/stress_inverse.log profiles/${pre}_stress_inverse.log
cp $O/e2e_profile_C.log profiles/${pre}_e2e_profile_C.log
tail -15 $O/pytest_gpu.log > profiles/${pre}_pytest_gpu_tail.log
cp $(find $O/prof_frob -name "*kernel_stats.csv" | head -1) profiles/${pre}_frob_kernel_stats.csv
cp $(find $O/prof_B_merged8 -name "*kernel_stats.csv" | head -1) profiles/${pre}_B_merged8_kernel_stats.csv
python - "$(find $O/prof_frob -name "*kernel_trace.csv" | head -1)" profiles/${pre}_frob_k_sweep_launches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('void k_sweep<')]
with open(sys.argv[2], 'w') as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs: the k_sweep rows of the SAME run the kernel_stats file summarises\n")
    out.write("kernel,start_ns,end_ns,duration_ms\n")
    d = []
    for r in rows:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        d.append(dur)
        out.write("%s,%s,%s,%.4f\n" % (r['Kernel_Name'], r['Start_Timestamp'], r['End_Timestamp'], dur))
print("k_sweep launches", len(d), "average %.3f ms (%.2f-%.2f)" % (sum(d) / len(d), min(d), max(d)))
PY
python tools/prof_summary.py calib $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE --out profiles/${pre}_fetch_calibration.json
python tools/prof_summary.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE --calib profiles/${pre}_fetch_calibration.json --out profiles/${pre}_pmc_update_traffic.json
python tools/prof_summary.py pmc $O/pmc_mfma --out profiles/${pre}_pmc_mfma.json | tail -1
python tools/prof_summary.py pmc $O/pmc_front $O/pmc_front2 --out profiles/${pre}_pmc_front.json | tail -1
python tools/kernel_resources.py > profiles/${pre}_kernel_resources.txt 2>/dev/null
python - $pre <<'PY'
import json, sys
pre = sys.argv[1]
for n in ('default', 'C_DI', 'B', 'B_merged8', 'B_merged8_grids0', 'D', 'E', 'E_p2', 'E_phased8', 'E_phased16', 'E_phased16_grids0'):
    try:
        x = json.loads(open('profiles/%s_bench_%s.json' % (pre, n)).read())
        print(n, 'value %.2f' % x['value'], 'ms/step %.3f' % x['ms_per_step'], 'frac %.3f' % x['roofline']['frac'], 'GHz %.3f' % x['roofline']['measured_shader_ghz'],
              'retries', x.get('sweep_retries'), 'e2e', x.get('end_to_end_gdca_sec'))
        if n == 'default':
            print('   other_configs', {k: (round(v['value'], 1), round(v.get('roofline', {}).get('frac', 0), 3)) for k, v in x['other_configs'].items()})
            print('   cpu_baseline sec/family', x['cpu_baseline'].get('sec_per_family'))
    except Exception as e:
        print(n, 'unreadable', e)
PY
