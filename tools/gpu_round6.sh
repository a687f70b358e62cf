#!/bin/bash
# Round 6: every measurement profiles/r06_* of the FINAL tree is made of, in one box call (re-runnable; ~20 min of GPU time):
#   /usr/local/graft/bin/gpurun --timeout 3400 -- 'bash tools/gpu_round6.sh r06'
# Output under gpurun_out/<tag>/; tools/prof_summary.py and the copy commands in profiles/README.md make the committed summaries.
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
timeout 600 python __graft_entry__.py smoke > $out/entry_smoke.log 2>&1 < /dev/null; echo "entry rc $?" >> $out/entry_smoke.log; tail -2 $out/entry_smoke.log
( timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -3 $out/pytest_gpu.log
# the driver's command (default flags: headline + other_configs incl. the whole E batch + cpu_baseline), then full-size lines of the other configurations
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null
timeout 600 python bench.py --score DI --no-other-configs > $out/bench_C_DI.json 2> $out/bench_C_DI.err < /dev/null
timeout 600 python bench.py --config B --no-cpu-baseline > $out/bench_B.json 2> $out/bench_B.err < /dev/null
timeout 600 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 > $out/bench_B_merged8.json 2> $out/bench_B_merged8.err < /dev/null
GDCA_PHASED_GRIDS=0 timeout 600 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 > $out/bench_B_merged8_grids0.json 2> $out/bench_B_merged8_grids0.err < /dev/null
timeout 900 python bench.py --config D > $out/bench_D.json 2> $out/bench_D.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline > $out/bench_E.json 2> $out/bench_E.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 2 > $out/bench_E_p2.json 2> $out/bench_E_p2.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 8 --phased > $out/bench_E_phased8.json 2> $out/bench_E_phased8.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 16 --phased > $out/bench_E_phased16.json 2> $out/bench_E_phased16.err < /dev/null
GDCA_PHASED_GRIDS=0 timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 16 --phased > $out/bench_E_phased16_grids0.json 2> $out/bench_E_phased16_grids0.err < /dev/null
for f in $out/bench_*.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'inv %.3f ms' % d['stage_ms']['ms_inverse'],
          'roofline %.1f TF (%.3f) at %.3f GHz' % (r['achieved'], r['frac'], r['measured_shader_ghz']), 'e2e', d.get('end_to_end_gdca_sec'),
          [(h['kernel'][:6], round(h['frac'], 3), h.get('traffic')) for h in d.get('roofline_hbm', [])],
          {k: round(v.get('value', 0), 2) for k, v in d.get('other_configs', {}).items()})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
timeout 600 python tools/stress_merged.py --rounds 40 --seed 11 > $out/stress_merged.log 2>&1 < /dev/null; tail -1 $out/stress_merged.log
timeout 600 python tools/stress_inverse.py > $out/stress_inverse.log 2>&1 < /dev/null; tail -1 $out/stress_inverse.log
timeout 200 python tools/e2e_profile.py C 5 > $out/e2e_profile_C.log 2>&1 < /dev/null; tail -3 $out/e2e_profile_C.log
# kernel-trace + stats of the driver's hot path: ONE run gives the per-kernel statistics AND the per-launch rows
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_frob -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_frob.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_B_merged8 -- python3 $R/bench.py --config B --pipeline 8 --phased --steps 40 --no-cpu-baseline > $R/$out/prof_B_merged8.log 2>&1 < /dev/null
# counters: separate passes, kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_$c.log 2>&1 < /dev/null
  [ -x $R/tools/_bin/ubench_fetch_calib ] && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/calib_$c -- $R/tools/_bin/ubench_fetch_calib > $R/$out/calib_$c.log 2>&1 < /dev/null
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_mfma.log 2>&1 < /dev/null
# what the front-end kernels do with their issue slots (k_hamming's missing 18 %, k_pair_tally)
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$out/pmc_front -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_front.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/$out/pmc_front2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_front2.log 2>&1 < /dev/null
cd $R
find $out -name "*.csv" -size +8M -delete
find $out -name "*agent_info*" -delete
du -sh $out
