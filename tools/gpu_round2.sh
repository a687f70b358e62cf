#!/bin/bash
# Second half of the round's measurements (counter passes, calibration, in-kernel traces, micro-benchmarks):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round2.sh r03'
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "schedule" > $out/pytest_schedules.log 2>&1; tail -2 $out/pytest_schedules.log )
timeout 600 python tools/stress_inverse.py > $out/stress_inverse.log 2>&1; tail -3 $out/stress_inverse.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/pmc_$c.log 2>&1
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/calib_$c -- $R/tools/_bin/ubench_fetch_calib > $R/$out/calib_$c.log 2>&1
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/pmc_mfma.log 2>&1
cd $R
GDCA_SWEEP_TRACE=$out/sweep_trace_C.txt timeout 300 python tools/sweep_trace.py 10000 > $out/sweep_trace_C.log 2>&1
GDCA_SWEEP_TRACE=$out/sweep_trace_D.txt timeout 300 python tools/sweep_trace.py 20000 > $out/sweep_trace_D.log 2>&1
GDCA_SWEEP_TRACE=$out/sweep_trace_B.txt timeout 300 python tools/sweep_trace.py 2560 5,6,7,8 > $out/sweep_trace_B.log 2>&1
GDCA_SWEEP_TRACE=$out/clock_ramp_C.txt timeout 300 python tools/clock_ramp.py 10000 8 > $out/clock_ramp_C.log 2>&1
# the pivot block alone (phase stamps) and the single-block chain options
[ -x tools/_bin/test_pivot ] && timeout 120 tools/_bin/test_pivot > $out/pivot_chain_phases.log 2>&1
timeout 900 bash tools/slab_ab.sh > $out/chain_slab_ab.log 2>&1
head -4 $out/sweep_trace_C.log
find $out -name "*agent_info*" -delete
# host side: feed rate of the batch driver and thread scaling of the CPU port (no GPU work)
timeout 900 bash tools/parse_bench.sh 256 /tmp/gdca_pb "1 16 32 64" > $out/parse_bench.log 2>&1
timeout 300 python tools/cpu_scaling.py > $out/cpu_scaling.log 2>&1
# (the dispatch micro-benchmarks of round 2 are tools/gpu_ubench.sh; the schedule sweep is tools/sweep_groups.py)
du -sh $out
