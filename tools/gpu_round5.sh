#!/bin/bash
# Round 5: every measurement profiles/r05_* is made of, in one call (re-runnable; ~20 min of GPU time):
#   /usr/local/graft/bin/gpurun --timeout 3400 -- 'bash tools/gpu_round5.sh r05'
# Output under gpurun_out/<tag>/; tools/prof_summary.py and the copy commands in profiles/README.md make the committed summaries.
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
timeout 600 python __graft_entry__.py smoke > $out/entry_smoke.log 2>&1 < /dev/null; echo "entry rc $?" >> $out/entry_smoke.log; tail -2 $out/entry_smoke.log
( timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
tail -3 $out/pytest_gpu.log
# the driver's command (default flags: headline + other_configs + cpu_baseline), then the full-size lines of the other configurations
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null
timeout 600 python bench.py --score DI --no-other-configs > $out/bench_C_DI.json 2> $out/bench_C_DI.err < /dev/null
timeout 600 python bench.py --config B --no-cpu-baseline > $out/bench_B.json 2> $out/bench_B.err < /dev/null
timeout 600 python bench.py --config B --no-cpu-baseline --pipeline 8 --phased --steps 80 > $out/bench_B_merged8.json 2> $out/bench_B_merged8.err < /dev/null
timeout 900 python bench.py --config D > $out/bench_D.json 2> $out/bench_D.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline > $out/bench_E.json 2> $out/bench_E.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 2 > $out/bench_E_p2.json 2> $out/bench_E_p2.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 8 --phased > $out/bench_E_phased8.json 2> $out/bench_E_phased8.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 16 --phased > $out/bench_E_phased16.json 2> $out/bench_E_phased16.err < /dev/null
timeout 900 python bench.py --config E --no-cpu-baseline --pipeline 4 --phased > $out/bench_E_phased4.json 2> $out/bench_E_phased4.err < /dev/null
for f in $out/bench_*.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'inv %.3f ms' % d['stage_ms']['ms_inverse'],
          'roofline %.1f TF (%.3f) at %.3f GHz' % (r['achieved'], r['frac'], r['measured_shader_ghz']), 'e2e', d.get('end_to_end_gdca_sec'),
          [(h['kernel'][:6], round(h['frac'], 3)) for h in d.get('roofline_hbm', [])])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
timeout 600 python tools/stress_merged.py --rounds 40 --seed 11 > $out/stress_merged.log 2>&1 < /dev/null; tail -1 $out/stress_merged.log
timeout 600 python tools/stress_inverse.py > $out/stress_inverse.log 2>&1 < /dev/null; tail -1 $out/stress_inverse.log
timeout 300 python tools/merge_probe.py --ks 1 4 8 > $out/merge_probe_B.log 2>&1 < /dev/null; tail -4 $out/merge_probe_B.log
timeout 300 python tools/merge_probe.py --sizes 200:20000 --theta -1 --ks 1 4 8 > $out/merge_probe_N200.log 2>&1 < /dev/null; tail -4 $out/merge_probe_N200.log
timeout 300 python tools/merge_probe.py --sizes 300:8000 --theta -1 --ks 1 2 4 > $out/merge_probe_N300.log 2>&1 < /dev/null; tail -4 $out/merge_probe_N300.log
timeout 200 python tools/e2e_profile.py C 5 > $out/e2e_profile_C.log 2>&1 < /dev/null; tail -3 $out/e2e_profile_C.log
INFLIGHT="2" timeout 600 bash tools/cli_batch_bench.sh 128 /tmp/gdca_cb > $out/cli_batch.log 2>&1 < /dev/null; tail -4 $out/cli_batch.log
( gaussdca.jl_amd/gdca_cli --batch /tmp/gdca_cb/in --out /tmp/gdca_cb/out2 --gpus 1 --inflight 2 --merge 8 --merge-blocks 57 2>&1 | tail -3 ) > $out/cli_batch_merge8.log 2>&1; cat $out/cli_batch_merge8.log
for n in 10000 20000; do
  GDCA_SWEEP_TRACE=$out/trace_$n.txt timeout 300 python tools/sweep_trace.py $n 99 > $out/trace_$n.log 2>&1
  grep "^# main\|^# shader" $out/trace_$n.log | cut -c1-330
done
rm -f $out/trace_*.txt
# kernel-trace + stats of the driver's hot path (profiled timings are not compared with un-profiled ones), both scores
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_frob -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_frob.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_di -- python3 $R/bench.py --score DI --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $R/$out/prof_di.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_B_merged8 -- python3 $R/bench.py --config B --pipeline 8 --phased --steps 40 --no-cpu-baseline > $R/$out/prof_B_merged8.log 2>&1 < /dev/null
# counters: separate passes, kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_$c.log 2>&1 < /dev/null
  [ -x $R/tools/_bin/ubench_fetch_calib ] && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/calib_$c -- $R/tools/_bin/ubench_fetch_calib > $R/$out/calib_$c.log 2>&1 < /dev/null
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $R/$out/pmc_mfma.log 2>&1 < /dev/null
cd $R
find $out -name "*.csv" -size +8M -delete
find $out -name "*agent_info*" -delete
du -sh $out
