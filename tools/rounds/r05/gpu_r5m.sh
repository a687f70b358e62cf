#!/bin/bash
# round 5, call m: hand-out order of the chain's items (GDCA_CHAIN_ORDER) -- parity of every forced schedule, then timings
out=gpurun_out/r5m; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "schedule or merged or side_by_side or inverse" > $out/pytest_sched.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_sched.log
timeout 600 python tools/option_probe.py 100,128,160,180,200,220,250,280,300,320,350,380,400,450,500,600 "CHAIN_ORDER=0;CHAIN_ORDER=1" 7 > $out/order_ab.log 2>&1; cat $out/order_ab.log
# chain compute units under the new order (single blocks: 8 from 28 blocks; groups of four: 16 / 12 / 10 / 6)
timeout 600 python tools/option_probe.py 180,200,250,300 "MCUS=6;MCUS=8;MCUS=10;MCUS=12;MCUS=16" 5 > $out/order_mcus_single.log 2>&1; cat $out/order_mcus_single.log
timeout 600 python tools/option_probe.py 380,400,450,500,600 "MCUS=4;MCUS=6;MCUS=8;MCUS=10;MCUS=12;MCUS=16" 5 > $out/order_mcus_multi.log 2>&1; cat $out/order_mcus_multi.log
# group size under the new order where the rule switches (47 .. 60 blocks)
timeout 600 python tools/option_probe.py 280,300,320,350,370 "GROUP=1;GROUP=2;GROUP=3;GROUP=4;GROUP=2,MCUS=8;GROUP=4,MCUS=8" 5 > $out/order_groups.log 2>&1; cat $out/order_groups.log
for n in 4000 6000; do
  GDCA_SWEEP_TRACE=$out/trace_$n.txt timeout 300 python tools/sweep_trace.py $n 5,6 > $out/trace_$n.log 2>&1
  grep "^# nblk\|^# main\|^# shader\|^# pivot\|^M list" $out/trace_$n.log | cut -c1-300
done
rm -f $out/trace_*.txt
