#!/bin/bash
# Round 4: every measurement profiles/r04_* is made of, in one call (re-runnable; ~12 min of GPU time):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_round4.sh r04'
# Output under gpurun_out/<tag>/; tools/prof_summary.py and the copy commands in profiles/README.md make the committed summaries.
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
timeout 600 python __graft_entry__.py smoke > $out/entry_smoke.log 2>&1 < /dev/null; echo "entry rc $?" >> $out/entry_smoke.log; tail -2 $out/entry_smoke.log
( timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log ) < /dev/null
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
for f in $out/bench_*.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'inv %.3f ms' % d['stage_ms']['ms_inverse'],
          'roofline %.1f TF (%.3f) at %.3f GHz' % (r['achieved'], r['frac'], r['measured_shader_ghz']), 'e2e', d.get('end_to_end_gdca_sec'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
# merged sweeps: probe (per-family inverse time by K and size), stress (random batches, poisoned workspaces)
timeout 300 python tools/merge_probe.py --ks 1 2 4 8 > $out/merge_probe_B.log 2>&1 < /dev/null
timeout 300 python tools/merge_probe.py --sizes 200:20000 --theta -1 --ks 2 4 8 > $out/merge_probe_N200.log 2>&1 < /dev/null
timeout 300 python tools/merge_probe.py --sizes 300:8000 --theta -1 --ks 2 4 > $out/merge_probe_N300.log 2>&1 < /dev/null
timeout 300 python tools/merge_probe.py --sizes 100:6000 160:9000 240:12000 300:8000 200:20000 130:7000 --theta -1 --ks 4 8 --tiles 2300 > $out/merge_probe_mixed.log 2>&1 < /dev/null
timeout 600 python tools/stress_merged.py --rounds 40 --seed 11 > $out/stress_merged.log 2>&1 < /dev/null; tail -1 $out/stress_merged.log
timeout 600 python tools/stress_inverse.py > $out/stress_inverse.log 2>&1 < /dev/null; tail -1 $out/stress_inverse.log
# conditioning (sweep alone / default path / LAPACK against refined columns)
timeout 1200 python -m pytest tests/test_gpu_conditioning.py -m gpu -q -s -p no:cacheprovider > $out/conditioning.log 2>&1 < /dev/null; grep "cond(C)\|passed\|failed" $out/conditioning.log | cut -c1-400
# the blocked Cholesky fallback against LAPACK (every inverse through it), status parity at tiny pseudocounts
PYTHONPATH=tests timeout 600 python tools/chol_probe.py --families > $out/chol_probe.txt 2>&1 < /dev/null; tail -1 $out/chol_probe.txt
# end to end, host feed, vmcnt ordering micro-benchmark
timeout 200 python tools/e2e_profile.py C 5 > $out/e2e_profile_C.log 2>&1 < /dev/null
timeout 200 python tools/e2e_profile.py D 3 > $out/e2e_profile_D.log 2>&1 < /dev/null
( cat /sys/fs/cgroup/cpu.max; nproc ) > $out/host_cpus.log 2>&1
PASSES=6 timeout 900 bash tools/parse_bench.sh 128 /tmp/gdca_pb "1 8 16 32 64" > $out/parse_bench.log 2>&1 < /dev/null
GDCA_FASTA_ZLIB=1 PASSES=6 timeout 600 bash tools/parse_bench.sh 128 /tmp/gdca_pb "1 16" > $out/parse_bench_zlib.log 2>&1 < /dev/null
timeout 300 bash tools/experiments/inflate/ab.sh > $out/inflate_ab.log 2>&1 < /dev/null
INFLIGHT="1 2 3" timeout 600 bash tools/cli_batch_bench.sh 128 /tmp/gdca_cb > $out/cli_batch.log 2>&1 < /dev/null
timeout 600 python bench.py --config E --families 128 --no-cpu-baseline > $out/bench_E128.json 2> $out/bench_E128.err < /dev/null   # the device-resident rate the CLI is compared with
[ -x tools/_bin/rank_time ] && ( tools/_bin/rank_time 500; tools/_bin/rank_time 1000 ) > $out/rank_time.log 2>&1 < /dev/null
[ -x tools/_bin/ubench_vmcnt_order ] && timeout 120 tools/_bin/ubench_vmcnt_order 1000 > $out/ubench_vmcnt_order.log 2>&1 < /dev/null
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
